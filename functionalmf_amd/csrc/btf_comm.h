// The ctx-owned communicator of a sharded run ("ctx owns device buffers, streams, RCCL communicators";
// 8(e): one all-gather of W after the W half-sweep, one of V after the V half-sweep, one all-reduced double for nu2).
//
// RCCL is bound at RUN time, never at link time: libbtf_hip.so has no DT_NEEDED entry for librccl (573 MB on disk;
// an unsharded run never maps it).  The first btf_comm_* call looks for a librccl.so.1 that the process has already
// mapped (dlopen RTLD_NOLOAD - e.g. the copy PyTorch bundles, which shares its SONAME with ROCm's: one RCCL, and through
// it one HIP runtime, per process, the same rule _native.load() follows for libamdhip64) and only then loads the one on
// the library search path (this library's RUNPATH is the ROCm lib directory).  BTF_RCCL_PATH overrides both.
#pragma once
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

}  // namespace btf
