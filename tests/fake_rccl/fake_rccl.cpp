// TEST INFRASTRUCTURE - not part of the product, never loaded unless MGF_RCCL_LIB names it.
//
// A stand-in for librccl that lets SEVERAL RANKS SHARE ONE GPU, so that the multi-rank half of the tile protocol under the C-ABI
// (mgf_amd/csrc/host_tiles_native.inc: count messages, ghost / velocity / migrant exchanges, the status agreement) runs for real -
// several processes, every send met by a receive of the same size - on a box with a single device, where RCCL itself refuses
// ("Duplicate GPU detected").  It implements exactly the entry points host_tiles_native.inc binds (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclAllReduce, ncclGetErrorString) with
// RCCL's matching rules, host-staged through POSIX shared memory:
//   * point-to-point messages between two ranks match IN ORDER of issue; a receive whose size differs from the matching send is an
//     ERROR here (RCCL would hang or corrupt memory) - so a protocol bug in the counts shows up as a failed test;
//   * operations inside ncclGroupStart / ncclGroupEnd progress together (sends do not wait for receives posted later in the group);
//     outside a group an operation completes on its own before the call returns - so a Send-before-Recv on both sides DEADLOCKS as it
//     would on RCCL, and the deadlock is reported after MGF_FAKE_RCCL_TIMEOUT_S (default 60) seconds instead of hanging the test run;
//   * ncclAllReduce: int32, ncclSum / ncclMax, up to 16 elements.
// Stream semantics: the stream is drained before a send buffer is read, and data is in the receive buffer when the call returns.
// Build: tests/fake_rccl/build.py (g++ against the HIP runtime).
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int kMaxRanks = 16;
constexpr size_t kPipeBytes = 8u << 20;  // per direction and pair of neighbouring ranks (a byte ring: messages larger than it stream through)

struct Pipe {
  std::atomic<uint64_t> w, r;  // bytes written / read so far
  uint8_t data[kPipeBytes];
};
struct Shared {
  std::atomic<uint32_t> joined;
  std::atomic<uint32_t> bar_count, bar_gen;
  int32_t red[kMaxRanks][16];
  Pipe pipe[kMaxRanks][2];  // pipe[src][0]: src -> src - 1, pipe[src][1]: src -> src + 1
};
struct Comm {
  Shared* sh = nullptr;
  int rank = 0, n = 1;
  std::string name;
};
struct Op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  hipStream_t stream;
  Comm* comm;
  std::vector<uint8_t> host;
  size_t done = 0;      // payload bytes moved
  bool header = false;  // the 8-byte length prefix moved
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local std::string g_err;

double timeout_s() {
  const char* e = getenv("MGF_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}
size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
Pipe* pipe_of(Comm* c, int src, int dst) { return &c->sh->pipe[src][dst > src ? 1 : 0]; }
size_t pipe_write(Pipe* p, const uint8_t* src, size_t n) {  // as much as fits; returns bytes written
  const uint64_t w = p->w.load(std::memory_order_relaxed), r = p->r.load(std::memory_order_acquire);
  const size_t room = kPipeBytes - (size_t)(w - r), k = n < room ? n : room;
  for (size_t i = 0; i < k;) {
    const size_t at = (size_t)((w + i) % kPipeBytes), run = std::min(k - i, kPipeBytes - at);
    memcpy(p->data + at, src + i, run);
    i += run;
  }
  p->w.store(w + k, std::memory_order_release);
  return k;
}
size_t pipe_read(Pipe* p, uint8_t* dst, size_t n) {
  const uint64_t r = p->r.load(std::memory_order_relaxed), w = p->w.load(std::memory_order_acquire);
  const size_t have = (size_t)(w - r), k = n < have ? n : have;
  for (size_t i = 0; i < k;) {
    const size_t at = (size_t)((r + i) % kPipeBytes), run = std::min(k - i, kPipeBytes - at);
    memcpy(dst + i, p->data + at, run);
    i += run;
  }
  p->r.store(r + k, std::memory_order_release);
  return k;
}
// Progress on a set of operations until all are complete.  Per (peer, direction) the operations run in issue order.
ncclResult_t run_ops(std::vector<Op>& ops) {
  // send buffers are read once the stream's earlier work is done; receive buffers are written once their earlier readers are
  for (Op& o : ops) {
    if (hipStreamSynchronize(o.stream) != hipSuccess) { g_err = "fake rccl: hipStreamSynchronize failed"; return ncclUnhandledCudaError; }
    break;  // (one stream per call site in host_tiles_native.inc)
  }
  for (Op& o : ops) {
    o.host.resize(o.bytes);
    if (o.send && o.bytes && hipMemcpy(o.host.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "fake rccl: device to host copy failed"; return ncclUnhandledCudaError; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true, moved = false;
    bool busy_send[kMaxRanks] = {false}, busy_recv[kMaxRanks] = {false};  // an earlier incomplete op to / from that peer blocks later ones
    for (Op& o : ops) {
      const bool complete = o.header && o.done == o.bytes;
      if (complete) continue;
      all = false;
      bool* busy = o.send ? busy_send : busy_recv;
      if (busy[o.peer]) continue;
      busy[o.peer] = true;
      Comm* c = o.comm;
      if (o.send) {
        Pipe* p = pipe_of(c, c->rank, o.peer);
        if (!o.header) {
          const uint64_t w = p->w.load(std::memory_order_relaxed), r = p->r.load(std::memory_order_acquire);
          if (kPipeBytes - (size_t)(w - r) < 8) continue;
          const uint64_t len = o.bytes;
          pipe_write(p, reinterpret_cast<const uint8_t*>(&len), 8);
          o.header = true; moved = true;
        }
        const size_t k = pipe_write(p, o.host.data() + o.done, o.bytes - o.done);
        o.done += k; moved = moved || k > 0;
      } else {
        Pipe* p = pipe_of(c, o.peer, c->rank);
        if (!o.header) {
          const uint64_t r = p->r.load(std::memory_order_relaxed), w = p->w.load(std::memory_order_acquire);
          if ((size_t)(w - r) < 8) continue;
          uint64_t len = 0;
          pipe_read(p, reinterpret_cast<uint8_t*>(&len), 8);
          if (len != o.bytes) {
            char b[256];
            snprintf(b, sizeof b, "fake rccl: rank %d posted a receive of %zu bytes from rank %d, the matching send carries %llu bytes", c->rank, o.bytes, o.peer, (unsigned long long)len);
            g_err = b;
            return ncclInvalidArgument;
          }
          o.header = true; moved = true;
        }
        const size_t k = pipe_read(p, o.host.data() + o.done, o.bytes - o.done);
        o.done += k; moved = moved || k > 0;
      }
    }
    if (all) break;
    if (!moved) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
        std::string s = "fake rccl: timeout (deadlock or missing peer); pending:";
        for (Op& o : ops) if (!(o.header && o.done == o.bytes)) { char b[96]; snprintf(b, sizeof b, " %s %zu B %s rank %d;", o.send ? "send" : "recv", o.bytes, o.send ? "to" : "from", o.peer); s += b; }
        g_err = s;
        return ncclSystemError;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }
  for (Op& o : ops)
    if (!o.send && o.bytes && hipMemcpy(o.buf, o.host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { g_err = "fake rccl: host to device copy failed"; return ncclUnhandledCudaError; }
  return ncclSuccess;
}
ncclResult_t barrier(Comm* c) {
  Shared* sh = c->sh;
  const uint32_t gen = sh->bar_gen.load(std::memory_order_acquire);
  if (sh->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->n) {
    sh->bar_count.store(0, std::memory_order_relaxed);
    sh->bar_gen.store(gen + 1, std::memory_order_release);
    return ncclSuccess;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (sh->bar_gen.load(std::memory_order_acquire) == gen) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { g_err = "fake rccl: timeout in a collective (a rank is missing)"; return ncclSystemError; }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return ncclSuccess;
}
ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t tb = type_bytes(type);
  if (!c || !tb || peer < 0 || peer >= c->n || peer == c->rank) { g_err = "fake rccl: bad argument"; return ncclInvalidArgument; }
  if (peer != c->rank - 1 && peer != c->rank + 1) { g_err = "fake rccl: only neighbouring ranks exchange messages in the tile protocol"; return ncclInvalidArgument; }
  Op o;
  o.send = send; o.buf = buf; o.bytes = count * tb; o.peer = peer; o.stream = stream; o.comm = c;
  if (g_depth > 0) { g_ops.push_back(std::move(o)); return ncclSuccess; }
  std::vector<Op> one;
  one.push_back(std::move(o));
  return run_ops(one);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, NCCL_UNIQUE_ID_BYTES, "/mgf_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)t);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) { g_err = "fake rccl: bad argument"; return ncclInvalidArgument; }
  id.internal[NCCL_UNIQUE_ID_BYTES - 1] = 0;
  Comm* c = new Comm();
  c->rank = rank; c->n = nranks; c->name = id.internal;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)sizeof(Shared)) != 0) { g_err = "fake rccl: shm_open / ftruncate failed"; delete c; return ncclSystemError; }
  void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);  // (fresh pages are zero: every counter starts at 0)
  close(fd);
  if (m == MAP_FAILED) { g_err = "fake rccl: mmap failed"; delete c; return ncclSystemError; }
  c->sh = reinterpret_cast<Shared*>(m);
  c->sh->joined.fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->joined.load(std::memory_order_acquire) < (uint32_t)nranks) {  // RCCL's init is a collective too
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { g_err = "fake rccl: timeout waiting for the other ranks to join"; return ncclSystemError; }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  shm_unlink(c->name.c_str());  // (the first rank to leave removes the name; the mappings live on)
  munmap(c->sh, sizeof(Shared));
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(true, const_cast<void*>(buf), count, type, peer, comm, stream);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(false, buf, count, type, peer, comm, stream);
}
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) { g_err = "fake rccl: ncclGroupEnd without ncclGroupStart"; return ncclInvalidUsage; }
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (ops.empty()) return ncclSuccess;
  return run_ops(ops);
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || type != ncclInt32 || count > 16 || (op != ncclSum && op != ncclMax)) { g_err = "fake rccl: this all-reduce is not implemented (int32, sum / max, <= 16 elements)"; return ncclInvalidArgument; }
  int32_t mine[16] = {0}, out[16];
  if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(mine, send, 4 * count, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "fake rccl: copy failed"; return ncclUnhandledCudaError; }
  memcpy(c->sh->red[c->rank], mine, sizeof(mine));
  ncclResult_t r = barrier(c);
  if (r != ncclSuccess) return r;
  for (size_t k = 0; k < count; ++k) {
    int32_t v = c->sh->red[0][k];
    for (int j = 1; j < c->n; ++j) v = op == ncclSum ? v + c->sh->red[j][k] : std::max(v, c->sh->red[j][k]);
    out[k] = v;
  }
  r = barrier(c);  // (nobody overwrites its slot before everybody has read)
  if (r != ncclSuccess) return r;
  if (hipMemcpy(recv, out, 4 * count, hipMemcpyHostToDevice) != hipSuccess) { g_err = "fake rccl: copy failed"; return ncclUnhandledCudaError; }
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "no error";
  return g_err.empty() ? "fake rccl: error" : g_err.c_str();
}
}
