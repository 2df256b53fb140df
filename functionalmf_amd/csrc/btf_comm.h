// The ctx-owned communicator of a sharded run ("ctx owns device buffers, streams, RCCL communicators";
// 8(e): one all-gather of W after the W half-sweep, one of V after the V half-sweep, one all-reduced double for nu2).
//
// RCCL is bound at RUN time, never at link time: libbtf_hip.so has no DT_NEEDED entry for librccl (573 MB on disk;
// an unsharded run never maps it).  The first btf_comm_* call looks for a librccl.so.1 that the process has already
// mapped (dlopen RTLD_NOLOAD - e.g. the copy PyTorch bundles, which shares its SONAME with ROCm's: one RCCL, and through
// it one HIP runtime, per process, the same rule _native.load() follows for libamdhip64) and only then loads the one on
// the library search path (this library's RUNPATH is the ROCm lib directory).  BTF_RCCL_PATH overrides both.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: every call goes through the table below
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace btf {

struct RcclApi {
  void* handle = nullptr;
  std::string error, origin;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;

  template <typename F>
  bool sym(F& f, const char* name) {
    f = reinterpret_cast<F>(dlsym(handle, name));
    if (!f) error = std::string("librccl: missing symbol ") + name;
    return f != nullptr;
  }
  bool open() {
    const char* forced = std::getenv("BTF_RCCL_PATH");
    if (forced && *forced) {
      handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
      origin = forced;
    } else {
      handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);      // the process's own copy, if it has one
      origin = "librccl.so.1 (already mapped)";
      if (!handle) { handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL); origin = "librccl.so.1 (search path)"; }
      if (!handle) { handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL); origin = "librccl.so (search path)"; }
    }
    if (!handle) {
      const char* e = dlerror();
      error = std::string("cannot load RCCL (") + origin + "): " + (e ? e : "unknown dlopen error");
      return false;
    }
    return sym(GetUniqueId, "ncclGetUniqueId") && sym(CommInitRank, "ncclCommInitRank") && sym(CommDestroy, "ncclCommDestroy") &&
           sym(CommCount, "ncclCommCount") && sym(AllGather, "ncclAllGather") && sym(AllReduce, "ncclAllReduce") &&
           sym(GroupStart, "ncclGroupStart") && sym(GroupEnd, "ncclGroupEnd") && sym(GetErrorString, "ncclGetErrorString") &&
           sym(GetVersion, "ncclGetVersion");
  }
};

// process-wide table, filled once (thread-safe); nullptr + *why when RCCL cannot be bound
inline RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static bool ok = false;
  static std::once_flag once;
  std::call_once(once, [] { ok = api.open(); });
  if (!ok) { if (why) *why = api.error; return nullptr; }
  return &api;
}

// Equal-chunk block decomposition of an axis: rank r owns [min(r*chunk, n), min((r+1)*chunk, n)), chunk = ceil(n / world) -
// the only decomposition an in-place all-gather with one send count reassembles (the tail ranks may be short or
// empty; W / V are padded by 64 rows / columns for them, btf_create).
inline int comm_chunk(int n, int world) { return (n + world - 1) / world; }
inline int comm_block_lo(int n, int rank, int world) { const long long lo = (long long)rank * comm_chunk(n, world); return (int)(lo < n ? lo : n); }
inline int comm_block_len(int n, int rank, int world) {
  const int lo = comm_block_lo(n, rank, world);
  const int c = comm_chunk(n, world);
  return (n - lo) < c ? (n - lo) : c;
}


// ---- the peer-window transport -----------------------------------------------------------------------------------------
// The same collectives without a communication library: every rank maps the W / V buffers and a small mailbox of every
// other rank (hipIpc handles - the ranks are processes of one node; or raw pointers when two contexts share a process)
// and ONE kernel per collective pushes this rank's block straight into the peers' buffers over xGMI (a store to a
// mapped peer address), flags it, and waits for the peers' blocks to land here.  Messages are 20 KB to 2.6 MB and the
// step around them is 40-110 us, so what counts is latency: one launch and two flag round trips, no ring, no proxy thread.
// Also the only device-side transport that lets several ranks share ONE GPU (RCCL refuses that), which is how the N > 1
// path of the C ABI is tested on a one-GPU box (tests/test_gpu_comm.py).
//
// Protocol of collective number e (every rank issues the same sequence of collectives, as with RCCL), for each peer r:
//   1. mailbox(r).arrive[me] = e     - this kernel started, so everything queued before it on my stream - every reader
//                                      of my W / V - is done: r may overwrite its block in my buffers
//   2. wait mailbox(me).arrive[r] >= e, then copy my block into r's buffer (and my all-reduce operands into
//      mailbox(r).red[me]), write-through stores, drained (s_waitcnt vmcnt(0)), and the last workgroup of the copy stores mailbox(r).done[me] = e
//   3. wait mailbox(me).done[r] >= e - r's block is here; the kernel ends, and with it the stream-ordered collective
// The mailbox is fine-grained device memory (flag loads and stores at system scope reach it past the L2s); W / V are
// ordinary allocations: the writer stores them write-through and drains before the flag, the readers are later kernels
// (the launch boundary invalidates their caches).  Every wait is bounded (BTF_PEER_TIMEOUT_MS, default 20 s): a dead peer turns
// into status code 3 and BTF_EHIP at the next check, never into a hung GPU (and once the status word is set, every
// later wait of this context returns at once).
// One process per rank is the deployment; several contexts of ONE process work too (raw pointers instead of handles)
// as long as their streams get a hardware queue each - the HIP runtime multiplexes a process's streams over 4
// (GPU_MAX_HW_QUEUES), and a waiting exchange kernel holds its queue.
constexpr int PEER_MAX = 64;
constexpr int PEER_RED = 16;
constexpr int PEER_THREADS = 256;
constexpr int PEER_DESC_BYTES = 256;          // BTF_PEER_DESC_BYTES of include/btf.h

struct PeerMailbox {
  unsigned long long arrive[PEER_MAX];
  unsigned long long done[PEER_MAX];
  double red[PEER_MAX][PEER_RED];
};
struct PeerDesc {                              // what a rank publishes: btf_peer_export
  hipIpcMemHandle_t W, V, box;
  unsigned long long pW, pV, pbox;             // the same three as raw pointers: valid inside the exporting process
  long long pid;
  int dev, magic;
  long long wbytes, vbytes;
};
static_assert(sizeof(PeerDesc) <= PEER_DESC_BYTES, "PeerDesc outgrew BTF_PEER_DESC_BYTES");
struct PeerTable {                             // device resident: where each rank's buffers are mapped HERE
  double* W[PEER_MAX];
  double* V[PEER_MAX];
  PeerMailbox* box[PEER_MAX];
};
struct PeerArgs {
  const PeerTable* tab;
  int rank, world, which, wpp;                 // which: 0 W, 1 V, 2 no block; wpp: workgroups per peer
  unsigned long long epoch;
  size_t off, len;                             // my block, in doubles
  const double* red_src; double* red_dst; int red_n;
  unsigned* counters;                          // [PEER_MAX + 1] zero between launches: per-peer copy counts, [PEER_MAX] operands read
  int* status;
  long long timeout_ticks;                     // of the 100 MHz wall clock
};

// Flags and payload are system-scope (sc0 sc1) accesses: stores are written through to the memory they target - a peer
// GPU's HBM over xGMI, or this GPU's - and acknowledged when they are there, so "s_waitcnt vmcnt(0)" is the whole release
// (no L2 write-back per workgroup: 8-byte stores, like the in-launch hand-offs of btf_fused.h); polls bypass the caches.
typedef __attribute__((address_space(1))) unsigned long long peer_u64;
__device__ __forceinline__ void peer_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store((peer_u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_store(double* p, double v) { peer_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ unsigned long long peer_load(const unsigned long long* p) {
  return __hip_atomic_load((const peer_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double peer_load(const double* p) { return __longlong_as_double((long long)peer_load(reinterpret_cast<const unsigned long long*>(p))); }
__device__ __forceinline__ void peer_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ inline bool peer_wait(const unsigned long long* p, unsigned long long e, long long ticks, int* status, int who) {
  const long long t0 = wall_clock64();
  while (peer_load(p) < e) {
    __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;      // already failed: drain
    if (wall_clock64() - t0 > ticks) {
      if (atomicCAS(status, 0, 3) == 0) status[1] = who;
      return false;
    }
  }
  return true;
}

__global__ __launch_bounds__(PEER_THREADS) void peer_exchange_kernel(PeerArgs a) {
  const int q = (int)blockIdx.x / a.wpp, part = (int)blockIdx.x % a.wpp, tid = (int)threadIdx.x;
  const int r = (a.rank + 1 + q) % a.world;                  // ranks start on different peers
  PeerMailbox* mine = a.tab->box[a.rank];
  PeerMailbox* theirs = a.tab->box[r];
  __shared__ int ok;
  if (tid == 0) {
    if (part == 0) peer_store(&theirs->arrive[a.rank], a.epoch);      // (everything this stream queued earlier is done: kernel boundary)
    ok = peer_wait(&mine->arrive[r], a.epoch, a.timeout_ticks, a.status, r) ? 1 : 0;
  }
  __syncthreads();
  const bool go = ok != 0;
  if (go) {
    if (a.len) {
      const double* src = (a.which == 0 ? a.tab->W[a.rank] : a.tab->V[a.rank]) + a.off;
      double* dst = (a.which == 0 ? a.tab->W[r] : a.tab->V[r]) + a.off;
      const size_t first = (size_t)part * PEER_THREADS + tid, stride = (size_t)a.wpp * PEER_THREADS;
      size_t i = first;
      for (; i + 7 * stride < a.len; i += 8 * stride) {      // eight loads in flight a thread
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) peer_store(dst + i + u * stride, v[u]);
      }
      for (; i < a.len; i += stride) peer_store(dst + i, src[i]);
    }
    if (a.red_n && part == 0 && tid < a.red_n) peer_store(&theirs->red[a.rank][tid], a.red_src[tid]);
  }
  peer_drain();
  __syncthreads();
  if (tid == 0) {
    if (a.red_n && part == 0) __hip_atomic_fetch_add(&a.counters[PEER_MAX], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned before = __hip_atomic_fetch_add(&a.counters[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == (unsigned)a.wpp - 1) {                     // every workgroup of this peer has drained its stores
      __hip_atomic_store(&a.counters[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (go) peer_store(&theirs->done[a.rank], a.epoch);
    }
    ok = go && peer_wait(&mine->done[r], a.epoch, a.timeout_ticks, a.status, r) ? 1 : 0;
  }
  __syncthreads();
  if (!a.red_n || blockIdx.x != 0) return;
  // the sum, in rank order on every rank (the same bits everywhere): workgroup 0, once every peer's operands are here and
  // every workgroup of this launch has read red_src (red_dst may be the same address)
  __shared__ int all;
  if (tid == 0) all = 1;
  __syncthreads();
  if (tid < a.world && tid != a.rank && !peer_wait(&mine->done[tid], a.epoch, a.timeout_ticks, a.status, tid)) all = 0;
  if (tid == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&a.counters[PEER_MAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(a.world - 1)) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeout_ticks) { all = 0; break; }
    }
    __hip_atomic_store(&a.counters[PEER_MAX], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid < a.red_n && all) {
    double s = 0.0;
    for (int p = 0; p < a.world; ++p) s += p == a.rank ? a.red_src[tid] : peer_load(&mine->red[p][tid]);
    a.red_dst[tid] = s;
  }
}

}  // namespace btf
