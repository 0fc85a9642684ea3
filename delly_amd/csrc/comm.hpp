// comm.hpp -- multi-GPU return path of the refinement results: junctions shard across the GPUs of a
// node with no data-path collective; what rank 0 needs for mergeSort / VCF emission (src/delly.h:149,179) is every
// rank's fixed-size result records PLUS the variable-length consensus / "REF,ALT" bytes (src/split.h:606-637).
// RCCL has no gatherv: ranks exchange their (record count, blob bytes) with one ncclAllGather, then the non-root ranks
// ncclSend and the root ncclRecv both pieces inside one group (point-to-point over xGMI; tens of KB to a few MB per
// rank, latency-bound).  RCCL is loaded with dlopen at the first use, so a single-GPU run never touches it and the
// library has no link-time dependency on it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the entry points are resolved with dlsym

#include <mutex>
#include <string>

namespace dh {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

inline RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // an RCCL the process already holds (e.g. the one PyTorch ships) wins: two copies would each build their own
    // topology over the same xGMI links
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) { api.error = std::string("dlopen(librccl): ") + dlerror(); return; }
    auto sym = [&](const char* name) { void* p = dlsym(api.lib, name); if (!p && api.error.empty()) api.error = std::string("dlsym ") + name; return p; };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return api;
}

// Cost-balanced assignment of junctions to ranks: predicted cost N^2 L^2 (all-pairs LCS / NW) +
// (N-1) L^2 c2 (progressive alignment) + m n (split alignment).  Longest-processing-time-first greedy: junctions by
// decreasing cost, each to the currently lightest rank.  owner[i] = rank of junction i; deterministic (ties by index).
inline void balance_by_cost(const double* cost, int n, int world, int32_t* owner) {
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  std::vector<double> load(std::max(world, 1), 0.0);
  for (int i : order) {
    int best = 0;
    for (int r = 1; r < world; ++r)
      if (load[r] < load[best]) best = r;
    owner[i] = best;
    load[best] += cost[i];
  }
}

}  // namespace dh
