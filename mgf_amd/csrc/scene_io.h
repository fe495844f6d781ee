// Scene I/O: the serde_json shape of the reference's persistent types -
//   BVH<AABB, usize> bvh.rs:29-47, Pool<T> / PoolEntry<T> pool.rs:25-41, AABB geom.rs:256-260, Mesh mesh.rs:31-37,
//   cgmath Point3 / Vector3 with the "serde" feature ({"x":..,"y":..,"z":..}).
// serde's defaults apply: structs are maps in declaration order, unit enum variants are strings, newtype / tuple /
// struct variants are single-key maps ("externally tagged"), Option::None is null, tuples are arrays.
//   BVH   {"root":R,"pool":{"len":L,"free_list":null|N,"entries":[E...]}}
//   E     "FreeListEnd" | {"FreeListPtr":{"next_free":N}} |
//         {"Occupied":{"height":H,"parent":P,"bounds":{"c":{x,y,z},"r":{x,y,z}},"node_type":{"Leaf":V}|{"Parent":[L,R]}}}
//   Mesh  {"x":{x,y,z},"verts":[{x,y,z}...],"faces":[[a,b,c]...],"bvh":BVH}
// Floats are written in their shortest round-trip form with serde_json's conventions ("1.0", "1e21", "1.5e-7").
// Host-only plumbing: no arithmetic happens here.
#pragma once
#include <charconv>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "host_bvh.h"

namespace mgf {
namespace sio {

inline void put_f32(std::string& o, float v) {
  char buf[48];
  auto r = std::to_chars(buf, buf + sizeof(buf), v);
  std::string s(buf, r.ptr);
  size_t e = s.find('e');
  std::string mant = e == std::string::npos ? s : s.substr(0, e), ex = e == std::string::npos ? "" : s.substr(e + 1);
  if (e == std::string::npos && mant.find('.') == std::string::npos && mant.find("inf") == std::string::npos && mant.find("nan") == std::string::npos)
    mant += ".0";
  o += mant;
  if (!ex.empty()) {  // "e+21" -> "e21", "e-07" -> "e-7"
    o += 'e';
    size_t k = 0;
    if (ex[k] == '+') ++k; else if (ex[k] == '-') { o += '-'; ++k; }
    while (k + 1 < ex.size() && ex[k] == '0') ++k;
    o += ex.substr(k);
  }
}
inline void put_u64(std::string& o, uint64_t v) { o += std::to_string(v); }
inline void put_v3(std::string& o, V3 v) { o += "{\"x\":"; put_f32(o, v.x); o += ",\"y\":"; put_f32(o, v.y); o += ",\"z\":"; put_f32(o, v.z); o += "}"; }

inline void write_bvh(std::string& o, const HostBvh& t) {
  o += "{\"root\":"; put_u64(o, t.root());
  o += ",\"pool\":{\"len\":"; put_u64(o, t.live());
  o += ",\"free_list\":";
  if (t.has_free()) put_u64(o, t.free_head()); else o += "null";
  o += ",\"entries\":[";
  for (uint64_t i = 0; i < t.slots(); ++i) {
    if (i) o += ',';
    const HostBvh::Node& n = t.node(i);
    if (n.state == HostBvh::kFreeEnd) { o += "\"FreeListEnd\""; continue; }
    if (n.state == HostBvh::kFreePtr) { o += "{\"FreeListPtr\":{\"next_free\":"; put_u64(o, n.next_free); o += "}}"; continue; }
    o += "{\"Occupied\":{\"height\":"; o += std::to_string(n.height);
    o += ",\"parent\":"; put_u64(o, n.parent);
    o += ",\"bounds\":{\"c\":"; put_v3(o, n.box.c); o += ",\"r\":"; put_v3(o, n.box.r); o += "}";
    o += ",\"node_type\":";
    if (n.leaf) { o += "{\"Leaf\":"; put_u64(o, n.value); o += "}"; }
    else { o += "{\"Parent\":["; put_u64(o, n.kid[0]); o += ','; put_u64(o, n.kid[1]); o += "]}"; }
    o += "}}";
  }
  o += "]}}";
}

// ---- a small JSON reader (objects keep their key order; numbers keep their text) ----
struct Val {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  std::string text;  // Num / Str
  std::vector<Val> arr;
  std::vector<std::pair<std::string, Val>> obj;
  const Val* get(const char* key) const {
    for (const auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
};
struct Parser {
  const char* p; const char* end; std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  bool str(std::string* out) {
    if (p >= end || *p != '"') return fail("expected a string");
    ++p; out->clear();
    while (p < end && *p != '"') {
      if (*p == '\\') { if (++p >= end) return fail("bad escape"); }
      out->push_back(*p++);
    }
    if (p >= end) return fail("unterminated string");
    ++p;
    return true;
  }
  bool value(Val* v, int depth = 0) {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= end) return fail("unexpected end of input");
    if (*p == '{') {
      v->kind = Val::Obj; ++p; ws();
      if (p < end && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!str(&k)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        ++p;
        Val c;
        if (!value(&c, depth + 1)) return false;
        v->obj.emplace_back(std::move(k), std::move(c));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v->kind = Val::Arr; ++p; ws();
      if (p < end && *p == ']') { ++p; return true; }
      for (;;) {
        Val c;
        if (!value(&c, depth + 1)) return false;
        v->arr.push_back(std::move(c));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v->kind = Val::Str; return str(&v->text); }
    if (end - p >= 4 && !strncmp(p, "null", 4)) { v->kind = Val::Null; p += 4; return true; }
    if (end - p >= 4 && !strncmp(p, "true", 4)) { v->kind = Val::Bool; v->b = true; p += 4; return true; }
    if (end - p >= 5 && !strncmp(p, "false", 5)) { v->kind = Val::Bool; v->b = false; p += 5; return true; }
    const char* s = p;
    while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) ++p;
    if (p == s) return fail("unexpected character");
    v->kind = Val::Num; v->text.assign(s, p);
    return true;
  }
};
inline bool as_u64(const Val* v, uint64_t* out) {
  if (!v || v->kind != Val::Num || v->text.empty() || v->text[0] == '-' || v->text.find_first_of(".eE") != std::string::npos) return false;
  *out = strtoull(v->text.c_str(), nullptr, 10);
  return true;
}
inline bool as_i32(const Val* v, int32_t* out) {
  if (!v || v->kind != Val::Num || v->text.find_first_of(".eE") != std::string::npos) return false;
  *out = (int32_t)strtol(v->text.c_str(), nullptr, 10);
  return true;
}
inline bool as_f32(const Val* v, float* out) {
  if (!v || v->kind != Val::Num) return false;
  *out = strtof(v->text.c_str(), nullptr);
  return true;
}
inline bool as_v3(const Val* v, V3* out) {
  return v && v->kind == Val::Obj && as_f32(v->get("x"), &out->x) && as_f32(v->get("y"), &out->y) && as_f32(v->get("z"), &out->z);
}
// BVH<AABB, usize> from its serde shape; validates the indices so that a damaged file cannot build a broken tree
inline bool read_bvh(const Val& v, HostBvh* t, std::string* err) {
  auto bad = [&](const char* m) { *err = m; return false; };
  const Val* pool = v.get("pool");
  uint64_t root = 0, len = 0;
  if (v.kind != Val::Obj || !pool || pool->kind != Val::Obj || !as_u64(v.get("root"), &root) || !as_u64(pool->get("len"), &len)) return bad("BVH: expected {root, pool:{len, free_list, entries}}");
  const Val* fl = pool->get("free_list");
  const Val* ent = pool->get("entries");
  if (!fl || !ent || ent->kind != Val::Arr) return bad("BVH: pool needs free_list and entries");
  bool has_free = fl->kind != Val::Null;
  uint64_t free_head = 0;
  if (has_free && !as_u64(fl, &free_head)) return bad("BVH: free_list must be null or an index");
  std::vector<HostBvh::Node> nodes(ent->arr.size());
  const uint64_t n = nodes.size();
  uint64_t occupied = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const Val& e = ent->arr[i];
    HostBvh::Node& nd = nodes[i];
    nd = HostBvh::Node();
    nd.state = HostBvh::kFreeEnd; nd.leaf = true; nd.height = -1; nd.parent = 0; nd.next_free = 0; nd.value = 0; nd.kid[0] = nd.kid[1] = 0;
    nd.box.c = mk3(0, 0, 0); nd.box.r = mk3(0, 0, 0);
    if (e.kind == Val::Str) { if (e.text != "FreeListEnd") return bad("PoolEntry: unknown unit variant"); continue; }
    if (e.kind != Val::Obj || e.obj.size() != 1) return bad("PoolEntry: expected a variant");
    const Val& body = e.obj[0].second;
    if (e.obj[0].first == "FreeListPtr") {
      nd.state = HostBvh::kFreePtr;
      if (!as_u64(body.get("next_free"), &nd.next_free) || nd.next_free >= n) return bad("FreeListPtr: bad next_free");
      continue;
    }
    if (e.obj[0].first != "Occupied") return bad("PoolEntry: unknown variant");
    nd.state = HostBvh::kUsed; ++occupied;
    const Val* b = body.get("bounds");
    const Val* nt = body.get("node_type");
    if (!as_i32(body.get("height"), &nd.height) || !as_u64(body.get("parent"), &nd.parent) || !b || !as_v3(b->get("c"), &nd.box.c) || !as_v3(b->get("r"), &nd.box.r) ||
        !nt || nt->kind != Val::Obj || nt->obj.size() != 1)
      return bad("BVHNode: expected {height, parent, bounds:{c, r}, node_type}");
    if (nt->obj[0].first == "Leaf") { nd.leaf = true; if (!as_u64(&nt->obj[0].second, &nd.value)) return bad("Leaf: expected an index"); }
    else if (nt->obj[0].first == "Parent") {
      const Val& pr = nt->obj[0].second;
      nd.leaf = false;
      if (pr.kind != Val::Arr || pr.arr.size() != 2 || !as_u64(&pr.arr[0], &nd.kid[0]) || !as_u64(&pr.arr[1], &nd.kid[1])) return bad("Parent: expected [left, right]");
    } else return bad("BVHNodeType: unknown variant");
  }
  if (occupied != len) return bad("Pool: len does not match the occupied entries");
  if (len && (root >= n || nodes[root].state != HostBvh::kUsed)) return bad("BVH: root is not an occupied entry");
  if (has_free && (free_head >= n || nodes[free_head].state == HostBvh::kUsed)) return bad("Pool: free_list points at an occupied entry");
  // a tree: every occupied entry but the root is the child of exactly one parent, the root of none (no cycle can then
  // be reached from the root: a walk down never meets an entry twice)
  std::vector<uint8_t> refs(n, 0);
  for (uint64_t i = 0; i < n; ++i) {
    const HostBvh::Node& nd = nodes[i];
    if (nd.state != HostBvh::kUsed) continue;
    if (nd.parent >= n && i != root) return bad("BVHNode: parent out of range");
    if (!nd.leaf) {
      if (nd.kid[0] == nd.kid[1]) return bad("BVHNode: both children are the same entry");
      for (int k = 0; k < 2; ++k) {
        if (nd.kid[k] >= n || nodes[nd.kid[k]].state != HostBvh::kUsed || nodes[nd.kid[k]].parent != i) return bad("BVHNode: child link does not match the child's parent");
        if (nd.kid[k] == root) return bad("BVHNode: the root is listed as a child");
        if (++refs[nd.kid[k]] > 1) return bad("BVHNode: an entry is the child of two parents");
      }
    }
  }
  for (uint64_t i = 0; i < n; ++i)
    if (nodes[i].state == HostBvh::kUsed && i != root && refs[i] != 1) return bad("BVHNode: an occupied entry has no parent entry");
  if (len) {  // ... and the walk from the root meets all of them (a detached ring of entries would pass the checks above)
    std::vector<uint64_t> stack(1, root);
    uint64_t seen = 0;
    while (!stack.empty()) {
      const uint64_t at = stack.back(); stack.pop_back();
      if (++seen > len) break;
      if (!nodes[at].leaf) { stack.push_back(nodes[at].kid[0]); stack.push_back(nodes[at].kid[1]); }
    }
    if (seen != len) return bad("BVH: the entries do not form one tree under the root");
  }
  // the free list visits each free entry at most once and ends at a FreeListEnd
  if (has_free) {
    uint64_t at = free_head, steps = 0;
    for (;;) {
      if (at >= n || nodes[at].state == HostBvh::kUsed) return bad("Pool: the free list runs into an occupied entry");
      if (nodes[at].state == HostBvh::kFreeEnd) break;
      if (++steps > n) return bad("Pool: the free list has a cycle");
      at = nodes[at].next_free;
    }
  }
  t->restore(std::move(nodes), root, len, has_free, free_head);
  return true;
}

}  // namespace sio
}  // namespace mgf
