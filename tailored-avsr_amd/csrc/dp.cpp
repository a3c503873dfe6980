// Data-parallel gradient exchange behind the C ABI: RCCL sum all-reduce of flat fp32 gradient buckets
// over xGMI, one communicator per process (= per GPU).  The reference is single-process: this is the only
// collective on the path.  RCCL is resolved at run time (dlopen): a process that already holds a copy (torch's
// librccl.so) shares it, and libtavsr_hip.so keeps loading on hosts without RCCL (single-GPU use).
//   rank 0: tavsr_dp_unique_id(id)  -> the caller ships the 128 bytes to every rank (torch.distributed store / broadcast)
//   all   : tavsr_dp_init(rank, nranks, id)   (collective: every rank calls it, on its own device)
//           tavsr_dp_allreduce(flat, n, stream)   in place, sum, enqueued on `stream` (no host sync)
//           tavsr_dp_broadcast(flat, n, root, stream)   parameter broadcast at start-up
//           tavsr_dp_destroy()
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_nranks = 0;

bool load_rccl() {
  if (g_rccl.h) return true;
  // an already loaded copy first (torch ships one), then the usual names
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
  if (!h) return false;
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Broadcast || !r.CommDestroy || !r.GetErrorString) return false;
  g_rccl = r;
  return true;
}

}  // namespace

#define TAVSR_NCCL(call, who)                                                              \
  do {                                                                                     \
    ncclResult_t r__ = (call);                                                             \
    if (r__ != ncclSuccess) {                                                              \
      ::tavsr::set_error("%s: RCCL error %d: %s", who, (int)r__, g_rccl.GetErrorString(r__)); \
      return 1000 + (int)r__;                                                              \
    }                                                                                      \
  } while (0)

extern "C" int tavsr_dp_unique_id(void* id128) {
  TAVSR_REQUIRE(id128 != nullptr, TAVSR_EINVAL, "dp_unique_id: null pointer");
  TAVSR_REQUIRE(load_rccl(), TAVSR_EUNSUPPORTED, "dp_unique_id: librccl.so not found (%s)", dlerror());
  static_assert(sizeof(ncclUniqueId) == 128, "tavsr_dp_unique_id ships 128 bytes");
  TAVSR_NCCL(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(id128)), "dp_unique_id");
  return TAVSR_OK;
}

extern "C" int tavsr_dp_init(int32_t rank, int32_t nranks, const void* id128) {
  TAVSR_REQUIRE(id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, TAVSR_EINVAL, "dp_init: bad rank %d of %d", rank,
                nranks);
  TAVSR_REQUIRE(g_comm == nullptr, TAVSR_EINVAL, "dp_init: already initialised (call tavsr_dp_destroy first)");
  TAVSR_REQUIRE(load_rccl(), TAVSR_EUNSUPPORTED, "dp_init: librccl.so not found (%s)", dlerror());
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  TAVSR_NCCL(g_rccl.CommInitRank(&g_comm, nranks, id, rank), "dp_init");
  g_nranks = nranks;
  return TAVSR_OK;
}

extern "C" int32_t tavsr_dp_world(void) { return g_comm ? g_nranks : 0; }

extern "C" int tavsr_dp_allreduce(float* flat, int64_t n, tavsr_stream_t stream) {
  TAVSR_REQUIRE(g_comm != nullptr, TAVSR_EINVAL, "dp_allreduce: tavsr_dp_init has not run");
  TAVSR_REQUIRE(flat != nullptr || n == 0, TAVSR_EINVAL, "dp_allreduce: null buffer");
  if (n <= 0) return TAVSR_OK;
  TAVSR_NCCL(g_rccl.AllReduce(flat, flat, (size_t)n, ncclFloat32, ncclSum, g_comm, static_cast<hipStream_t>(stream)), "dp_allreduce");
  return TAVSR_OK;
}

extern "C" int tavsr_dp_broadcast(float* flat, int64_t n, int32_t root, tavsr_stream_t stream) {
  TAVSR_REQUIRE(g_comm != nullptr, TAVSR_EINVAL, "dp_broadcast: tavsr_dp_init has not run");
  TAVSR_REQUIRE((flat != nullptr || n == 0) && root >= 0 && root < g_nranks, TAVSR_EINVAL, "dp_broadcast: bad arguments");
  if (n <= 0) return TAVSR_OK;
  TAVSR_NCCL(g_rccl.Broadcast(flat, flat, (size_t)n, ncclFloat32, root, g_comm, static_cast<hipStream_t>(stream)), "dp_broadcast");
  return TAVSR_OK;
}

extern "C" int tavsr_dp_destroy(void) {
  if (g_comm) {
    TAVSR_NCCL(g_rccl.CommDestroy(g_comm), "dp_destroy");
    g_comm = nullptr;
    g_nranks = 0;
  }
  return TAVSR_OK;
}
