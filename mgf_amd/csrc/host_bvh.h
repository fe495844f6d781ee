// Host-side dynamic AABB tree behind mgf_bvh_* and Mesh.
//
// Mirrors the reference's BVH<AABB, usize> (src/bvh.rs:30-480) over Pool<T> (src/pool.rs:26-113):
// SAH-descent insert, sibling-promoting remove, AVL-style single rotations, LIFO reuse of freed
// node ids.  Insert/remove are inherently sequential pointer surgery, so they live on the host;
// the tree is flattened to `DevNode`s and traversed on the GPU (kernels.h: bvh_traverse), which
// keeps the reference's stack discipline so hits arrive in the same DFS order.
#pragma once
#include <algorithm>
#include <vector>

#include "dev_geom.h"

namespace mgf {

// 32-byte device node: c.xyz + word0, r.xyz + word1.
//   leaf:   word0 = 0x80000000 | (value & 0x7fffffff), word1 = 0
//   parent: word0 = child1, word1 = child2
struct DevNode { float cx, cy, cz; uint32_t w0; float rx, ry, rz; uint32_t w1; };

class HostBvh {
 public:
  enum : uint8_t { kFreeEnd = 0, kFreePtr = 1, kUsed = 2 };
  struct Node {
    uint8_t state;       // Pool entry state (pool.rs:26-32)
    bool leaf;
    int32_t height;      // -1 for fresh nodes (bvh.rs:118)
    uint64_t parent;
    uint64_t next_free;  // FreeListPtr
    uint64_t value;      // Leaf(V)
    uint64_t kid[2];     // Parent(l, r)
    Box box;
  };

  bool empty() const { return live_ == 0; }
  uint64_t live() const { return live_; }
  uint64_t slots() const { return nodes_.size(); }
  uint64_t root() const { return root_; }
  bool used(uint64_t i) const { return i < nodes_.size() && nodes_[i].state == kUsed; }
  const Node& node(uint64_t i) const { return nodes_[i]; }
  void reserve(uint64_t n) { nodes_.reserve(n); }
  void clear() { nodes_.clear(); live_ = 0; has_free_ = false; root_ = 0; ++version_; }
  uint64_t version() const { return version_; }
  bool has_free() const { return has_free_; }        // Pool::free_list (pool.rs:38) is Some(free_head())
  uint64_t free_head() const { return free_head_; }
  // deserialised state (scene_io.h has validated the links)
  void restore(std::vector<Node> nodes, uint64_t root, uint64_t live, bool has_free, uint64_t free_head) {
    nodes_ = std::move(nodes); root_ = root; live_ = live; has_free_ = has_free; free_head_ = free_head; ++version_;
  }

  // BVH::insert bvh.rs:125-217
  uint64_t insert(const Box& b, uint64_t value) {
    ++version_;
    uint64_t leaf = alloc(b, true, value, 0, 0);
    if (live_ == 1) { root_ = leaf; return leaf; }
    uint64_t best = root_;
    while (!nodes_[best].leaf) {
      const Node& nb = nodes_[best];
      float area = box_area(nb.box);
      float comb_area = box_area(box_combine(nb.box, b));
      float stay = comb_area * 2.0f;
      float inherit = (comb_area - area) * 2.0f;
      float cost[2];
      for (int k = 0; k < 2; ++k) {
        const Node& ch = nodes_[nb.kid[k]];
        float merged = box_area(box_combine(b, ch.box));
        cost[k] = ch.leaf ? (merged + inherit) : (merged - box_area(ch.box) + inherit);
      }
      if (stay < cost[0] && stay < cost[1]) break;
      best = cost[0] < cost[1] ? nb.kid[0] : nb.kid[1];
    }
    uint64_t old_parent = nodes_[best].parent;
    uint64_t np = alloc(box_combine(b, nodes_[best].box), false, 0, best, leaf);
    nodes_[np].parent = old_parent;
    nodes_[np].height = nodes_[best].height + 1;
    if (best != root_) relink(old_parent, best, np);
    else root_ = np;
    nodes_[best].parent = np;
    nodes_[leaf].parent = np;
    for (uint64_t i = np;;) {
      i = balance(i);
      if (!nodes_[i].leaf) {
        refit(i);
        if (i == root_) break;
      }
      i = nodes_[i].parent;
    }
    return leaf;
  }

  // BVH::remove bvh.rs:220-260 (caller checks used(leaf))
  void remove(uint64_t leaf) {
    ++version_;
    uint64_t parent = nodes_[leaf].parent;
    release(leaf);
    if (leaf == root_) { root_ = 0; return; }
    if (nodes_[parent].leaf) return;
    uint64_t sib = nodes_[parent].kid[0] == leaf ? nodes_[parent].kid[1] : nodes_[parent].kid[0];
    if (root_ == parent) { root_ = sib; release(parent); return; }
    uint64_t gp = nodes_[parent].parent;
    relink(gp, parent, sib);
    nodes_[sib].parent = gp;
    release(parent);
    for (uint64_t i = gp;;) {
      i = balance(i);
      refit_remove(i);
      if (root_ == i) break;
      i = nodes_[i].parent;
    }
  }

  // BVH::query bvh.rs:283-310 on the host: every leaf whose bounds overlap `arg` (arg.overlaps(node), collision.rs:22-29), in the
  // reference's visiting order (push lchild, push rchild, pop).
  template <class F>
  void query(const Box& arg, F&& on_leaf) const {
    if (empty()) return;
    std::vector<uint64_t> stack{root_};
    while (!stack.empty()) {
      const uint64_t top = stack.back(); stack.pop_back();
      const Node& n = nodes_[top];
      if (!box_overlaps(arg, n.box)) continue;
      if (n.leaf) on_leaf(n.value);
      else { stack.push_back(n.kid[0]); stack.push_back(n.kid[1]); }
    }
  }

  // Leaves in the order a full BVH::query visits them (bvh.rs:283-310: push lchild, push rchild, pop rchild first).
  // Any query reports its hits in this order, whatever it prunes - so a hit list can be found by other means and
  // sorted by rank.  rank_of_value[v] for leaf value v (values must be dense indices), leaf_of_value[v] = node id.
  void dfs_ranks(size_t n_values, std::vector<uint32_t>* rank_of_value, std::vector<uint32_t>* value_of_rank,
                 std::vector<uint32_t>* leaf_of_value, std::vector<uint32_t>* parent_of_node) const {
    rank_of_value->assign(n_values, 0u); value_of_rank->clear(); leaf_of_value->assign(n_values, 0u);
    parent_of_node->assign(nodes_.size(), 0u);
    for (size_t i = 0; i < nodes_.size(); ++i) parent_of_node->at(i) = (uint32_t)nodes_[i].parent;
    if (empty()) return;
    std::vector<uint64_t> stack{root_};
    while (!stack.empty()) {
      uint64_t top = stack.back(); stack.pop_back();
      const Node& n = nodes_[top];
      if (n.leaf) {
        if (n.value < n_values) { (*rank_of_value)[n.value] = (uint32_t)value_of_rank->size(); (*leaf_of_value)[n.value] = (uint32_t)top; }
        value_of_rank->push_back((uint32_t)n.value);
      } else { stack.push_back(n.kid[0]); stack.push_back(n.kid[1]); }
    }
  }

  // Flatten for the device.  Unused slots become empty leaves that nothing points at.
  void flatten(std::vector<DevNode>* out) const {
    out->resize(nodes_.size());
    for (size_t i = 0; i < nodes_.size(); ++i) {
      const Node& n = nodes_[i];
      DevNode d;
      d.cx = n.box.c.x; d.cy = n.box.c.y; d.cz = n.box.c.z;
      d.rx = n.box.r.x; d.ry = n.box.r.y; d.rz = n.box.r.z;
      if (n.state != kUsed || n.leaf) { d.w0 = 0x80000000u | (uint32_t)(n.value & 0x7fffffffu); d.w1 = 0; }
      else { d.w0 = (uint32_t)n.kid[0]; d.w1 = (uint32_t)n.kid[1]; }
      (*out)[i] = d;
    }
  }

 private:
  std::vector<Node> nodes_;
  uint64_t live_ = 0, root_ = 0, free_head_ = 0, version_ = 0;
  bool has_free_ = false;

  // Pool::push pool.rs:81-96 — reuse the most recently freed slot first.
  uint64_t alloc(const Box& b, bool leaf, uint64_t value, uint64_t k0, uint64_t k1) {
    Node n;
    n.state = kUsed; n.leaf = leaf; n.height = -1; n.parent = 0; n.next_free = 0; n.value = value;
    n.kid[0] = k0; n.kid[1] = k1; n.box = b;
    ++live_;
    if (has_free_) {
      uint64_t slot = free_head_;
      if (nodes_[slot].state == kFreeEnd) has_free_ = false;
      else free_head_ = nodes_[slot].next_free;
      nodes_[slot] = n;
      return slot;
    }
    nodes_.push_back(n);
    return nodes_.size() - 1;
  }
  // Pool::remove pool.rs:100-113
  void release(uint64_t i) {
    nodes_[i].state = has_free_ ? kFreePtr : kFreeEnd;
    nodes_[i].next_free = free_head_;
    free_head_ = i;
    has_free_ = true;
    --live_;
  }
  void relink(uint64_t parent, uint64_t from, uint64_t to) {
    Node& p = nodes_[parent];
    if (p.leaf) return;
    if (p.kid[0] == from) p.kid[0] = to;
    else p.kid[1] = to;
  }
  void refit(uint64_t i) {  // insert order: height then bounds (bvh.rs:204-207)
    Node& n = nodes_[i];
    n.height = 1 + std::max(nodes_[n.kid[0]].height, nodes_[n.kid[1]].height);
    n.box = box_combine(nodes_[n.kid[0]].box, nodes_[n.kid[1]].box);
  }
  void refit_remove(uint64_t i) { refit(i); }

  // One AVL-style rotation lifting `up` (a child of `a`) above `a`.  bvh.rs:371-480.
  // side = 1: up is kid[1] (the "c" case :378-427); side = 0: up is kid[0] (the "b" case :428-477).
  uint64_t rotate(uint64_t a, int side) {
    uint64_t up = nodes_[a].kid[side];
    uint64_t other = nodes_[a].kid[1 - side];
    if (nodes_[up].leaf) return up;
    uint64_t g0 = nodes_[up].kid[0], g1 = nodes_[up].kid[1];
    nodes_[up].parent = nodes_[a].parent;
    nodes_[a].parent = up;
    if (root_ == a) root_ = up;
    else relink(nodes_[up].parent, a, up);
    // The taller grandchild stays under `up`; the other one moves under `a`.
    uint64_t stay = nodes_[g0].height > nodes_[g1].height ? g0 : g1;
    uint64_t move = nodes_[g0].height > nodes_[g1].height ? g1 : g0;
    nodes_[up].kid[0] = a;
    nodes_[up].kid[1] = stay;
    if (side == 1) { nodes_[a].kid[0] = other; nodes_[a].kid[1] = move; }  // Parent(b, g|f)
    else { nodes_[a].kid[0] = move; nodes_[a].kid[1] = other; }            // Parent(e|d, c)
    nodes_[move].parent = a;
    // bounds: combine(other, move) in both cases (:404,:416 use (b, g|f); :454,:466 use (c, e|d))
    nodes_[a].box = box_combine(nodes_[other].box, nodes_[move].box);
    nodes_[up].box = box_combine(nodes_[a].box, nodes_[stay].box);
    nodes_[a].height = 1 + std::max(nodes_[other].height, nodes_[move].height);
    nodes_[up].height = 1 + std::max(nodes_[a].height, nodes_[stay].height);
    return up;
  }
  uint64_t balance(uint64_t a) {
    if (nodes_[a].height < 2 || nodes_[a].leaf) return a;
    uint64_t b = nodes_[a].kid[0], c = nodes_[a].kid[1];
    if (nodes_[c].height > nodes_[b].height + 1) return rotate(a, 1);
    if (nodes_[b].height > nodes_[c].height + 1) return rotate(a, 0);
    return a;
  }
};

}  // namespace mgf
