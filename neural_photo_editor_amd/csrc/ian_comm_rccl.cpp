// ian_comm_rccl: the collective table of the training step (ian_comm_ops, include/ian_train.h) filled from librccl directly --
// what a C / C++ caller of ian_train_step uses for the data-parallel step (north_star: RCCL gradient all-reduce over xGMI,
// one process per GPU).  The Python host fills the same table from torch.distributed (trainer.Comm.ops), whose "nccl" backend
// IS RCCL; this file is the torch-free route.  RCCL collectives are stream-ordered:
//   allreduce_sum -> ncclAllReduce(in place, float32, sum) on the stream the trainer hands over (its side stream),
//   wait_all      -> an event recorded behind the last all-reduce, the given (compute) stream waits for it on the device,
//   allgather     -> ncclAllGather on the given stream.
// librccl is resolved with dlopen at the first call, not linked: libian.so loads (and every single-GPU path runs) on hosts
// without it, and a process that already maps an RCCL (PyTorch-ROCm ships one) reuses that copy.
// The 128-byte unique id travels out of band (whatever the launcher has: a file, MPI, a torch.distributed store): rank 0 calls
// ian_rccl_unique_id, every rank calls ian_rccl_comm_create with the same bytes.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <string>

#include "../../include/ian_train.h"

namespace {

thread_local std::string g_rccl_err;

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
};

Api* api() {
  static Api a;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // the soname first: a process that already maps an RCCL under it (PyTorch-ROCm) gets that copy back; RTLD_LOCAL: our handle
    // only, nothing of it enters the global symbol scope
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (a.lib) {
      a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
      a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
      a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
      a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
      if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.AllReduce || !a.AllGather) a.lib = nullptr;
    }
  }
  return a.lib ? &a : nullptr;
}

struct Ctx {
  ncclComm_t comm = nullptr;
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;   // stream of the most recent allreduce_sum since the last wait_all
  bool pending = false;
};

int fail(const char* what, ncclResult_t r) {
  Api* a = api();
  g_rccl_err = std::string(what) + ": " + (a ? a->GetErrorString(r) : "librccl not loaded");
  return 1;
}

int cb_allreduce(void* c, float* buf, int64_t count, void* stream) {
  Ctx* x = (Ctx*)c;
  Api* a = api();
  if (!x || !a || !buf || count <= 0) return 1;
  const ncclResult_t r = a->AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, x->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllReduce", r);
  x->last = (hipStream_t)stream;
  x->pending = true;
  return 0;
}
int cb_wait_all(void* c, void* stream) {
  Ctx* x = (Ctx*)c;
  if (!x) return 1;
  if (!x->pending) return 0;
  x->pending = false;
  if (x->last == (hipStream_t)stream) return 0;            // same stream: already ordered
  if (hipEventRecord(x->ev, x->last) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, x->ev, 0) != hipSuccess) {
    g_rccl_err = "hipEventRecord / hipStreamWaitEvent failed in wait_all";
    (void)hipGetLastError();
    return 1;
  }
  return 0;
}
int cb_allgather(void* c, const float* src, float* dst, int64_t count, void* stream) {
  Ctx* x = (Ctx*)c;
  Api* a = api();
  if (!x || !a || !src || !dst || count <= 0) return 1;
  const ncclResult_t r = a->AllGather(src, dst, (size_t)count, ncclFloat, x->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail("ncclAllGather", r);
  return 0;
}

}  // namespace

extern "C" {

const char* ian_rccl_last_error(void) { return g_rccl_err.c_str(); }

/* rank 0: 128 bytes to hand to every rank (ncclGetUniqueId).  -10: librccl could not be loaded. */
int ian_rccl_unique_id(void* out128) {
  Api* a = api();
  if (!out128) return -1;
  if (!a) {
    g_rccl_err = "librccl.so could not be loaded (dlopen)";
    return -10;
  }
  ncclUniqueId id;
  const ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return -fail("ncclGetUniqueId", r);
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

/* every rank, on its own GPU (hipSetDevice before): joins the communicator and fills *ops for ian_trainer_set_comm.  Blocks until
   all `world` ranks have called it (ncclCommInitRank). */
int ian_rccl_comm_create(const void* id128, int32_t rank, int32_t world, ian_comm_ops* ops) {
  Api* a = api();
  if (!id128 || !ops || world < 1 || rank < 0 || rank >= world) return -1;
  if (!a) {
    g_rccl_err = "librccl.so could not be loaded (dlopen)";
    return -10;
  }
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  Ctx* x = new Ctx();
  const ncclResult_t r = a->CommInitRank(&x->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete x;
    return -fail("ncclCommInitRank", r);
  }
  if (hipEventCreateWithFlags(&x->ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    (void)a->CommDestroy(x->comm);
    delete x;
    g_rccl_err = "hipEventCreateWithFlags failed";
    return -20;
  }
  memset(ops, 0, sizeof *ops);
  ops->world = world;
  ops->rank = rank;
  ops->ctx = x;
  ops->allreduce_sum = cb_allreduce;
  ops->wait_all = cb_wait_all;
  ops->allgather = cb_allgather;
  return 0;
}

/* after ian_trainer_destroy of every trainer that used the table */
void ian_rccl_comm_destroy(ian_comm_ops* ops) {
  if (!ops || !ops->ctx) return;
  Ctx* x = (Ctx*)ops->ctx;
  Api* a = api();
  if (x->ev) (void)hipEventDestroy(x->ev);
  if (a && x->comm) (void)a->CommDestroy(x->comm);
  delete x;
  ops->ctx = nullptr;
}

}  // extern "C"
