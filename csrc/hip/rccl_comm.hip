// Native RCCL communicator (replaces what tf.train.Server's gRPC runtime gives the reference,
// /root/reference/demo2/train.py:13-29): one ncclComm_t per process, created from a ncclUniqueId that rank 0 publishes through
// OUR rendezvous store (csrc/host/tcp_store.cc) — torch.distributed is not involved.  The collectives are plain enqueues on
// the caller's HIP stream, so the step executor captures them into the step's hipGraph next to its kernels: the RCCL
// data-parallel step is ONE graph launch (it used to be three graph launches with host-issued collectives in between).
//
// RCCL is bound at run time (dlopen): the process usually has one loaded already (PyTorch ships its own librccl), and two
// copies of the library in one process is what linking a second one at build time would risk; it also keeps libdtf_hip.so
// loadable on boxes without RCCL.  Only the stable public C API is used.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclInProgress = 7 };
// ncclDataType_t (nccl.h: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8, bfloat16 9)
enum { ncclInt8 = 0, ncclFloat32 = 7, ncclBfloat16 = 9 };
enum { ncclSum = 0, ncclMax = 2 };       // ncclRedOp_t (sum 0, prod 1, max 2, min 3)

struct Api {
  void* lib;
  int (*GetVersion)(int*);
  int (*GetUniqueId)(ncclUniqueId*);
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  int (*CommDestroy)(ncclComm_t);
  int (*CommAbort)(ncclComm_t);
  int (*CommGetAsyncError)(ncclComm_t, int*);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  const char* (*GetErrorString)(int);
};
Api g_api{};
char g_err[256] = "";

template <class F>
bool sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(g_api.lib, name));
  if (!f) snprintf(g_err, sizeof g_err, "RCCL symbol %s not found", name);
  return f != nullptr;
}

// hint: optional path of a librccl (e.g. the one inside the torch wheel); an already loaded one wins
int load_api(const char* hint) {
  if (g_api.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", hint, "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !g_api.lib; ++pass)
    for (const char* n : names) {
      if (!n || !*n) continue;
      g_api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_api.lib) break;
    }
  if (!g_api.lib) {
    snprintf(g_err, sizeof g_err, "librccl not found (%s)", dlerror());
    return -2;
  }
  const bool ok = sym(g_api.GetVersion, "ncclGetVersion") && sym(g_api.GetUniqueId, "ncclGetUniqueId") &&
                  sym(g_api.CommInitRank, "ncclCommInitRank") && sym(g_api.CommDestroy, "ncclCommDestroy") &&
                  sym(g_api.CommAbort, "ncclCommAbort") && sym(g_api.CommGetAsyncError, "ncclCommGetAsyncError") &&
                  sym(g_api.AllReduce, "ncclAllReduce") && sym(g_api.Broadcast, "ncclBroadcast") &&
                  sym(g_api.AllGather, "ncclAllGather") && sym(g_api.GroupStart, "ncclGroupStart") &&
                  sym(g_api.GroupEnd, "ncclGroupEnd") && sym(g_api.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    g_api = Api{};
    return -3;
  }
  return 0;
}

struct DtfRccl {
  ncclComm_t comm;
  int rank, world;
};

int fail(int rc, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
  return 1000 + rc;
}
}  // namespace

DTF_API const char* dtf_rccl_last_error() { return g_err; }

// 0 on success; *version = NCCL_VERSION_CODE of the library that was bound
DTF_API int dtf_rccl_load(const char* hint, int* version) {
  const int rc = load_api(hint);
  if (rc) return rc;
  if (version) g_api.GetVersion(version);
  return 0;
}

// rank 0: fills out[128] with a fresh id (to be published through the rendezvous store)
DTF_API int dtf_rccl_unique_id(void* out128) {
  if (load_api(nullptr)) return -2;
  ncclUniqueId id;
  const int rc = g_api.GetUniqueId(&id);
  if (rc != ncclSuccess) return fail(rc, "ncclGetUniqueId");
  memcpy(out128, &id, sizeof id);
  return 0;
}

// collective over `world` processes, each with its own current HIP device
DTF_API void* dtf_rccl_init(const void* id128, int rank, int world) {
  if (load_api(nullptr) || rank < 0 || rank >= world) return nullptr;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  DtfRccl* h = new DtfRccl{nullptr, rank, world};
  const int rc = g_api.CommInitRank(&h->comm, world, id, rank);
  if (rc != ncclSuccess) {
    fail(rc, "ncclCommInitRank");
    delete h;
    return nullptr;
  }
  return h;
}

DTF_API int dtf_rccl_rank(const void* h) { return static_cast<const DtfRccl*>(h)->rank; }
DTF_API int dtf_rccl_world(const void* h) { return static_cast<const DtfRccl*>(h)->world; }

// recv[i] = sum / max over ranks of send[i] (fp32, n elements; in place when send == recv); enqueued on st (capturable)
DTF_API int dtf_rccl_allreduce_f32(void* hv, const float* send, float* recv, long n, int op_max, hipStream_t st) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h || n <= 0) return -22;
  const int rc = g_api.AllReduce(send, recv, (size_t)n, ncclFloat32, op_max ? ncclMax : ncclSum, h->comm, st);
  return rc == ncclSuccess ? 0 : fail(rc, "ncclAllReduce");
}

// in-place bf16 sum over ranks (n elements): half the wire bytes of the fp32 form, for gradient buckets whose consumer keeps
// fp32 master weights (the Inception fine-tune: 87 MB -> 44 MB per step)
DTF_API int dtf_rccl_allreduce_bf16(void* hv, void* buf, long n, hipStream_t st) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h || n <= 0) return -22;
  const int rc = g_api.AllReduce(buf, buf, (size_t)n, ncclBfloat16, ncclSum, h->comm, st);
  return rc == ncclSuccess ? 0 : fail(rc, "ncclAllReduce (bf16)");
}

// several in-place fp32 sums as one RCCL group (one launch): the gradient buckets of a step
DTF_API int dtf_rccl_allreduce_f32_multi(void* hv, float* const* bufs, const long* counts, int nbuf, hipStream_t st) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h || nbuf <= 0) return -22;
  int rc = g_api.GroupStart();
  for (int i = 0; i < nbuf && rc == ncclSuccess; ++i)
    rc = g_api.AllReduce(bufs[i], bufs[i], (size_t)counts[i], ncclFloat32, ncclSum, h->comm, st);
  const int rc2 = g_api.GroupEnd();
  if (rc != ncclSuccess) return fail(rc, "ncclAllReduce (group)");
  return rc2 == ncclSuccess ? 0 : fail(rc2, "ncclGroupEnd");
}

DTF_API int dtf_rccl_broadcast(void* hv, void* buf, long bytes, int root, hipStream_t st) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h || bytes <= 0 || root < 0 || root >= h->world) return -22;
  const int rc = g_api.Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, h->comm, st);
  return rc == ncclSuccess ? 0 : fail(rc, "ncclBroadcast");
}

// recv[r*bytes .. (r+1)*bytes) = rank r's send[0 .. bytes)
DTF_API int dtf_rccl_allgather(void* hv, const void* send, void* recv, long bytes, hipStream_t st) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h || bytes <= 0) return -22;
  const int rc = g_api.AllGather(send, recv, (size_t)bytes, ncclInt8, h->comm, st);
  return rc == ncclSuccess ? 0 : fail(rc, "ncclAllGather");
}

// 0 = healthy; otherwise 1000 + the asynchronous ncclResult_t of the communicator (a peer died, a transport failed)
DTF_API int dtf_rccl_async_error(void* hv) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h) return -22;
  int st = ncclSuccess;
  const int rc = g_api.CommGetAsyncError(h->comm, &st);
  if (rc != ncclSuccess) return fail(rc, "ncclCommGetAsyncError");
  return (st == ncclSuccess || st == ncclInProgress) ? 0 : fail(st, "asynchronous RCCL error");
}

DTF_API void dtf_rccl_destroy(void* hv, int abort) {
  DtfRccl* h = static_cast<DtfRccl*>(hv);
  if (!h) return;
  if (abort) g_api.CommAbort(h->comm); else g_api.CommDestroy(h->comm);
  delete h;
}
