// ORACLE — TEST INFRASTRUCTURE ONLY (see mgf_math.hpp header).
// CPU restatement of src/pool.rs:26-113,153-173 (Pool<T>, LIFO free list) and
// src/bvh.rs:30-47,86-310,371-480 (dynamic AVL-balanced AABB tree: SAH insert,
// remove, balance, stack-DFS query).  Pinned by pool.rs:255-389 and bvh.rs:514-529.
#pragma once
#include <stdexcept>
#include <vector>

#include "mgf_geom.hpp"

namespace mgfo {

// Pool<T> pool.rs:37-113.  Entry states: FreeListEnd, FreeListPtr{next}, Occupied.
template <class T>
struct Pool {
  enum State : uint8_t { FREE_END = 0, FREE_PTR = 1, OCCUPIED = 2 };
  struct Entry { State st; size_t next_free; T item; };
  size_t len = 0;
  bool has_free = false;
  size_t free_list = 0;
  std::vector<Entry> entries;

  bool empty() const { return len == 0; }
  void clear() { len = 0; has_free = false; entries.clear(); }
  size_t push(const T& item) {  // pool.rs:81-96
    len += 1;
    if (has_free) {
      size_t free_item = free_list;
      Entry& e = entries[free_item];
      if (e.st == FREE_END) has_free = false;
      else { has_free = true; free_list = e.next_free; }
      e.st = OCCUPIED;
      e.item = item;
      return free_item;
    }
    size_t i = entries.size();
    entries.push_back(Entry{OCCUPIED, 0, item});
    return i;
  }
  T remove(size_t i) {  // pool.rs:100-113
    Entry& e = entries[i];
    if (e.st != OCCUPIED) throw std::runtime_error("pool index is not occupied");
    T item = e.item;
    if (has_free) { e.st = FREE_PTR; e.next_free = free_list; }
    else { e.st = FREE_END; }
    has_free = true;
    free_list = i;
    len -= 1;
    return item;
  }
  T& operator[](size_t i) {  // pool.rs:153-173
    if (i >= entries.size() || entries[i].st != OCCUPIED) throw std::runtime_error("pool index is not occupied");
    return entries[i].item;
  }
  const T& operator[](size_t i) const {
    if (i >= entries.size() || entries[i].st != OCCUPIED) throw std::runtime_error("pool index is not occupied");
    return entries[i].item;
  }
  bool occupied(size_t i) const { return i < entries.size() && entries[i].st == OCCUPIED; }
};

// BVH<AABB, V> bvh.rs:30-47
template <class V>
struct BVH {
  struct Node {
    int32_t height;
    size_t parent;
    AABB bounds;
    bool is_leaf;
    V leaf;
    size_t child1, child2;
  };
  size_t root = 0;
  Pool<Node> pool;

  bool empty() const { return pool.empty(); }
  void clear() { root = 0; pool.clear(); }
  const AABB& operator[](size_t i) const { return pool[i].bounds; }  // bvh.rs:483-492
  size_t get_root() const {
    if (empty()) throw std::runtime_error("BVH is empty, there is no root node");
    return root;
  }
  const V& get_leaf(size_t i) const {
    if (!pool[i].is_leaf) throw std::runtime_error("node is not a leaf");
    return pool[i].leaf;
  }

  size_t insert_node(const AABB& b, bool is_leaf, const V& val, size_t c1, size_t c2) {  // bvh.rs:114-121
    return pool.push(Node{-1, 0, b, is_leaf, val, c1, c2});
  }

  // bvh.rs:125-217
  size_t insert(const AABB& bounds, const V& val) {
    size_t leaf = insert_node(bounds, true, val, 0, 0);
    if (pool.len == 1) { root = leaf; return leaf; }
    size_t best = root;
    for (;;) {
      if (!pool[best].is_leaf) {
        size_t child1 = pool[best].child1, child2 = pool[best].child2;
        AABB curr_bounds = pool[best].bounds;
        float area = aabb_surface_area(curr_bounds);
        AABB combined_bounds = aabb_combine(curr_bounds, bounds);
        float combined_area = aabb_surface_area(combined_bounds);
        float no_descent_cost = combined_area * 2.0f;
        float inheritance_cost = (combined_area - area) * 2.0f;
        auto child_cost = [&](size_t child) -> float {
          if (!pool[child].is_leaf) {
            float old_area = aabb_surface_area(pool[child].bounds);
            float new_area = aabb_surface_area(aabb_combine(bounds, pool[child].bounds));
            return new_area - old_area + inheritance_cost;
          }
          return aabb_surface_area(aabb_combine(bounds, pool[child].bounds)) + inheritance_cost;
        };
        float child1_cost = child_cost(child1);
        float child2_cost = child_cost(child2);
        if (no_descent_cost < child1_cost && no_descent_cost < child2_cost) break;
        best = child1_cost < child2_cost ? child1 : child2;
      } else {
        break;
      }
    }
    size_t old_parent = pool[best].parent;
    AABB best_bounds = pool[best].bounds;
    size_t new_parent = insert_node(aabb_combine(bounds, best_bounds), false, V{}, best, leaf);
    pool[new_parent].parent = old_parent;
    pool[new_parent].height = pool[best].height + 1;
    if (best != root) {
      Node& op = pool[old_parent];
      if (!op.is_leaf) {
        if (op.child1 == best) op.child1 = new_parent;
        else op.child2 = new_parent;
      }
    } else {
      root = new_parent;
    }
    pool[best].parent = new_parent;
    pool[leaf].parent = new_parent;
    size_t i = pool[leaf].parent;
    for (;;) {
      i = balance(i);
      if (!pool[i].is_leaf) {
        size_t c1 = pool[i].child1, c2 = pool[i].child2;
        pool[i].height = 1 + std::max(pool[c1].height, pool[c2].height);
        pool[i].bounds = aabb_combine(pool[c1].bounds, pool[c2].bounds);
        if (i == root) break;
      }
      i = pool[i].parent;
    }
    return leaf;
  }

  // bvh.rs:220-260
  void remove(size_t leaf) {
    size_t parent = pool[leaf].parent;
    pool.remove(leaf);
    if (leaf == root) { root = 0; return; }
    if (!pool[parent].is_leaf) {
      size_t child1 = pool[parent].child1, child2 = pool[parent].child2;
      size_t sibling = child1 == leaf ? child2 : child1;
      if (root != parent) {
        size_t grand_parent = pool[parent].parent;
        Node& gp = pool[grand_parent];
        if (!gp.is_leaf) {
          if (gp.child1 == parent) gp.child1 = sibling;
          else gp.child2 = sibling;
        }
        pool[sibling].parent = grand_parent;
        pool.remove(parent);
        size_t i = grand_parent;
        for (;;) {
          i = balance(i);
          if (!pool[i].is_leaf) {
            size_t c1 = pool[i].child1, c2 = pool[i].child2;
            pool[i].bounds = aabb_combine(pool[c1].bounds, pool[c2].bounds);
            pool[i].height = 1 + std::max(pool[c1].height, pool[c2].height);
            if (root == i) break;
            i = pool[i].parent;
          }
          // NOTE: the reference loops forever if pool[i] is a leaf here; it cannot
          // be, since i is always an ancestor of `sibling`.
        }
      } else {
        root = sibling;
        pool.remove(parent);
      }
    }
  }

  // bvh.rs:345-369 raytrace: `hit(bounds, &inter)` is Intersects<AABB> of the caller's particle (collision.rs:202-236);
  // cb(value, Intersection with the leaf's bounds).  Same stack discipline as query.
  template <class T, class F>
  void raytrace(T&& hit, F&& cb) const {
    if (empty()) return;
    size_t inl[64];
    std::vector<size_t> spill;
    size_t sp = 0;
    auto push = [&](size_t v) { if (sp < 64) inl[sp] = v; else spill.push_back(v); ++sp; };
    auto pop = [&]() -> size_t { --sp; if (sp < 64) return inl[sp]; size_t v = spill.back(); spill.pop_back(); return v; };
    push(root);
    while (sp > 0) {
      size_t top = pop();
      const Node& n = pool[top];
      decltype(hit(n.bounds)) inter = hit(n.bounds);
      if (inter.first) {
        if (n.is_leaf) cb(n.leaf, inter.second);
        else { push(n.child1); push(n.child2); }
      }
    }
  }
  // bvh.rs:283-310: explicit stack, push lchild then rchild, pop rchild first.
  template <class F>
  void query(const AABB& arg_bounds, F&& cb) const {
    if (empty()) return;
    // SmallVec<[usize; 64]> bvh.rs:294 — inline 64, heap spill beyond.
    size_t inl[64];
    std::vector<size_t> spill;
    size_t sp = 0;
    auto push = [&](size_t v) { if (sp < 64) inl[sp] = v; else spill.push_back(v); ++sp; };
    auto pop = [&]() -> size_t { --sp; if (sp < 64) return inl[sp]; size_t v = spill.back(); spill.pop_back(); return v; };
    push(root);
    while (sp > 0) {
      size_t top = pop();
      const Node& n = pool[top];
      if (aabb_overlaps(arg_bounds, n.bounds)) {
        if (n.is_leaf) cb(n.leaf);
        else { push(n.child1); push(n.child2); }
      }
    }
  }

  // bvh.rs:371-480
  size_t balance(size_t a) {
    if (pool[a].height < 2) return a;
    if (!pool[a].is_leaf) {
      size_t b = pool[a].child1, c = pool[a].child2;
      if (pool[c].height > pool[b].height + 1) {
        if (!pool[c].is_leaf) {
          size_t f = pool[c].child1, g = pool[c].child2;
          pool[c].parent = pool[a].parent;
          pool[a].parent = c;
          if (root == a) {
            root = c;
          } else {
            size_t parent = pool[c].parent;
            Node& pn = pool[parent];
            if (!pn.is_leaf) {
              if (pn.child1 == a) pn.child1 = c;
              else pn.child2 = c;
            }
          }
          if (pool[f].height > pool[g].height) {
            pool[c].child1 = a; pool[c].child2 = f;
            pool[a].child1 = b; pool[a].child2 = g;
            pool[g].parent = a;
            pool[a].bounds = aabb_combine(pool[b].bounds, pool[g].bounds);
            pool[c].bounds = aabb_combine(pool[a].bounds, pool[f].bounds);
            pool[a].height = 1 + std::max(pool[b].height, pool[g].height);
            pool[c].height = 1 + std::max(pool[a].height, pool[f].height);
          } else {
            pool[c].child1 = a; pool[c].child2 = g;
            pool[a].child1 = b; pool[a].child2 = f;
            pool[f].parent = a;
            pool[a].bounds = aabb_combine(pool[b].bounds, pool[f].bounds);
            pool[c].bounds = aabb_combine(pool[a].bounds, pool[g].bounds);
            pool[a].height = 1 + std::max(pool[b].height, pool[f].height);
            pool[c].height = 1 + std::max(pool[a].height, pool[g].height);
          }
        }
        return c;
      }
      if (pool[b].height > pool[c].height + 1) {
        if (!pool[b].is_leaf) {
          size_t d = pool[b].child1, e = pool[b].child2;
          pool[b].parent = pool[a].parent;
          pool[a].parent = b;
          if (root == a) {
            root = b;
          } else {
            size_t parent = pool[b].parent;
            Node& pn = pool[parent];
            if (!pn.is_leaf) {
              if (pn.child1 == a) pn.child1 = b;
              else pn.child2 = b;
            }
          }
          if (pool[d].height > pool[e].height) {
            pool[b].child1 = a; pool[b].child2 = d;
            pool[a].child1 = e; pool[a].child2 = c;
            pool[e].parent = a;
            pool[a].bounds = aabb_combine(pool[c].bounds, pool[e].bounds);
            pool[b].bounds = aabb_combine(pool[a].bounds, pool[d].bounds);
            pool[a].height = 1 + std::max(pool[c].height, pool[e].height);
            pool[b].height = 1 + std::max(pool[a].height, pool[d].height);
          } else {
            pool[b].child1 = a; pool[b].child2 = e;
            pool[a].child1 = d; pool[a].child2 = c;
            pool[d].parent = a;
            pool[a].bounds = aabb_combine(pool[c].bounds, pool[d].bounds);
            pool[b].bounds = aabb_combine(pool[a].bounds, pool[e].bounds);
            pool[a].height = 1 + std::max(pool[c].height, pool[d].height);
            pool[b].height = 1 + std::max(pool[a].height, pool[e].height);
          }
        }
        return b;
      }
    }
    return a;
  }
};

}  // namespace mgfo
