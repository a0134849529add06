// tq_comm.cpp — the one exchange step of the segment-per-GPU deployment: an RCCL
// all-gather of the per-segment top-k lists, behind the C ABI so that a Rust host can call it.
//
// replaces: the fan-in of Searcher::search_with_executor (src/core/searcher.rs:230-235: the
// executor maps collect_segment over the segment readers and hands the fruits to merge_fruits,
// src/core/executor.rs:61-104) for segments that live on different GPUs.  Every rank contributes
// the [rows][stride] result slabs of its local segments; the gathered layout
// [rank][rows][stride] is exactly the [segment][query][stride] input of tq_merge_topk_device
// when rank r holds the segments r*S .. r*S+S-1.
//
// librccl is opened at run time (dlopen): a process that already carries an RCCL (PyTorch ships
// its own librccl.so.1) shares that copy instead of loading a second one, and hosts that never
// go multi-GPU do not need the library at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Hosts without the RCCL development headers still build the library (it only dlopen()s librccl):
// the handful of types the seven entry points use, as rccl.h / nccl.h declare them.
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
  char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclFloat32 = 7 } ncclDataType_t;
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/tantivy_amd.h"
#include "tq_launch.h"

static_assert(TQ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "tq_comm id size follows ncclUniqueId");

namespace {

struct Rccl {
  void *handle = nullptr;
  std::string origin;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

void load_rccl() {
  const char *env = getenv("TQ_RCCL_LIB");
  const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      g_rccl_error = dlerror();
      continue;
    }
    Rccl r;
    r.handle = h;
    r.origin = n;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GroupStart &&
        r.GroupEnd && r.GetErrorString) {
      g_rccl = r;
      return;
    }
    g_rccl_error = std::string(n) + ": RCCL entry points missing";
    dlclose(h);
  }
}

int need_rccl(const char *where) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.handle)
    return tq_internal_fail(TQ_ERR_UNSUPPORTED, where,
                            ("librccl not loadable: " + g_rccl_error).c_str());
  return TQ_OK;
}

int nccl_fail(const char *where, ncclResult_t r) {
  return tq_internal_fail(TQ_ERR_HIP, where, g_rccl.GetErrorString(r));
}

}  // namespace

struct tq_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
};

extern "C" {

int tq_comm_unique_id(uint8_t *id_out) {
  if (!id_out) return tq_internal_fail(TQ_ERR_INVALID, "tq_comm_unique_id", "null argument");
  int rc = need_rccl("tq_comm_unique_id");
  if (rc != TQ_OK) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  memcpy(id_out, id.internal, TQ_COMM_ID_BYTES);
  return TQ_OK;
}

int tq_comm_init(tq_ctx *ctx, int device, const uint8_t *id, int rank, int world, tq_comm **out) {
  if (!ctx || !id || !out) return tq_internal_fail(TQ_ERR_INVALID, "tq_comm_init", "null argument");
  if (world < 1 || rank < 0 || rank >= world)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_comm_init", "rank outside 0..world-1");
  if (!tq_internal_ctx_has_device(ctx, device))
    return tq_internal_fail(TQ_ERR_INVALID, "tq_comm_init", "device not part of this context");
  int rc = need_rccl("tq_comm_init");
  if (rc != TQ_OK) return rc;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return tq_internal_fail(TQ_ERR_HIP, "hipSetDevice", hipGetErrorString(e));
  ncclUniqueId uid;
  memcpy(uid.internal, id, TQ_COMM_ID_BYTES);
  tq_comm *c = new tq_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return TQ_OK;
}

void tq_comm_free(tq_comm *c) {
  if (!c) return;
  if (c->comm && g_rccl.handle) {
    (void)hipSetDevice(c->device);
    (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
}

int tq_comm_info(const tq_comm *c, int *rank, int *world, const char **library) {
  if (!c) return tq_internal_fail(TQ_ERR_INVALID, "tq_comm_info", "null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (library) *library = g_rccl.origin.c_str();
  return TQ_OK;
}

int tq_allgather_topk(tq_comm *c, const float *d_scores, const uint32_t *d_docs,
                      const uint32_t *d_counts, uint32_t n_rows, uint32_t stride,
                      float *d_all_scores, uint32_t *d_all_docs, uint32_t *d_all_counts,
                      void *hip_stream) {
  if (!c || !d_scores || !d_docs || !d_counts || !d_all_scores || !d_all_docs || !d_all_counts)
    return tq_internal_fail(TQ_ERR_INVALID, "tq_allgather_topk", "null argument");
  if (n_rows == 0) return TQ_OK;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return tq_internal_fail(TQ_ERR_HIP, "hipSetDevice", hipGetErrorString(e));
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t cells = (size_t)n_rows * stride;
  // one grouped collective: the three arrays travel in the same RCCL launch, each landing in its
  // [rank][rows][stride] slab (rank order = segment order)
  ncclResult_t r = g_rccl.GroupStart();
  if (r == ncclSuccess) r = g_rccl.AllGather(d_scores, d_all_scores, cells, ncclFloat32, c->comm, st);
  if (r == ncclSuccess) r = g_rccl.AllGather(d_docs, d_all_docs, cells, ncclUint32, c->comm, st);
  if (r == ncclSuccess) r = g_rccl.AllGather(d_counts, d_all_counts, n_rows, ncclUint32, c->comm, st);
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
  if (r2 != ncclSuccess) return nccl_fail("ncclGroupEnd", r2);
  return TQ_OK;
}

}  // extern "C"
