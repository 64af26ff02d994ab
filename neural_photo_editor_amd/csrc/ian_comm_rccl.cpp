// ian_comm_rccl: the collective table of the training step (ian_comm_ops, include/ian_train.h) filled from librccl directly --
// what a C / C++ caller of ian_train_step uses for the data-parallel step (north_star: RCCL gradient all-reduce over xGMI,
// one process per GPU), and since round 5 what bench.py / train_cli.py use at N > 1 (trainer.NativeRcclComm; the
// torch.distributed filler trainer.Comm.ops is the fallback).  RCCL collectives are stream-ordered:
//   allreduce_sum -> ncclAllReduce(in place, float32, sum) on the stream the trainer hands over (its side stream),
//   wait_all      -> one event per DISTINCT stream an all-reduce was issued on since the last wait, the given (compute) stream
//                    waits for each of them on the device,
//   allgather     -> ncclAllGather on the given stream, on a SECOND communicator when one was added (ian_rccl_comm_add_gather):
//                    one communicator serialises its collectives in issue order whatever streams they are on, so a 16 MB gradient
//                    bucket handed over during backward would sit in front of the next batch-statistics all-gather the compute
//                    stream blocks on; with its own communicator the small all-gather overtakes the bucket.
// librccl is resolved with dlopen at the first call, not linked, and NOTHING of <rccl/rccl.h> is needed to build this file (the
// seven prototypes below are RCCL's stable NCCL-compatible C ABI): libian.so builds and loads -- and every single-GPU path runs
// -- on hosts without RCCL, and a process that already maps an RCCL (PyTorch-ROCm ships one) reuses that copy.
// The 128-byte unique ids travel out of band (whatever the launcher has: a file, MPI, a torch.distributed store): rank 0 calls
// ian_rccl_unique_id (once per communicator), every rank calls ian_rccl_comm_create [+ ian_rccl_comm_add_gather] with the same bytes.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ian_train.h"

namespace {

// ---- the slice of the NCCL / RCCL C ABI this file calls (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE,
// ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0) ------------------------------------------------------------------------
constexpr int kIdBytes = 128;
struct RcclUniqueId { char internal[kIdBytes]; };
typedef struct ncclComm* RcclComm;
typedef int RcclResult;
constexpr int kRcclSuccess = 0, kRcclFloat32 = 7, kRcclSum = 0;

thread_local std::string g_rccl_err;

struct Api {
  void* lib = nullptr;
  RcclResult (*GetUniqueId)(RcclUniqueId*) = nullptr;
  RcclResult (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  RcclResult (*CommDestroy)(RcclComm) = nullptr;
  const char* (*GetErrorString)(RcclResult) = nullptr;
  RcclResult (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  RcclResult (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
};

Api* api() {
  static Api a;
  static std::once_flag once;   // two threads creating communicators at the same time resolve the library once
  std::call_once(once, [] {
    // the soname first: a process that already maps an RCCL under it (PyTorch-ROCm) gets that copy back; RTLD_LOCAL: our handle
    // only, nothing of it enters the global symbol scope
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (!a.lib) return;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.AllReduce || !a.AllGather) a.lib = nullptr;
  });
  return a.lib ? &a : nullptr;
}

struct Ctx {
  RcclComm comm = nullptr;      // gradient all-reduces
  RcclComm gather = nullptr;    // batch-statistics / MinibatchLayer all-gathers (nullptr: they share `comm`)
  int world = 1, rank = 0;
  // streams an all-reduce was issued on since the last wait_all, each with its own event (a caller with two side streams gets
  // both waited for; the trainer uses one)
  std::vector<std::pair<hipStream_t, hipEvent_t>> pending;
  std::vector<hipEvent_t> free_events;
};

int fail(const char* what, RcclResult r) {
  Api* a = api();
  g_rccl_err = std::string(what) + ": " + (a ? a->GetErrorString(r) : "librccl not loaded");
  return 1;
}

int cb_allreduce(void* c, float* buf, int64_t count, void* stream) {
  Ctx* x = (Ctx*)c;
  Api* a = api();
  if (!x || !a || !buf || count <= 0) return 1;
  const hipStream_t st = (hipStream_t)stream;
  // the event wait_all will need for this stream is secured BEFORE the collective is issued: a failure here must not leave an
  // all-reduce in flight that nothing tracks
  bool tracked = false;
  for (auto& p : x->pending)
    if (p.first == st) tracked = true;
  if (!tracked) {
    hipEvent_t e = nullptr;
    if (!x->free_events.empty()) {
      e = x->free_events.back();
      x->free_events.pop_back();
    } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      g_rccl_err = "hipEventCreateWithFlags failed in allreduce_sum (nothing was issued)";
      return 1;
    }
    x->pending.push_back({st, e});
  }
  const RcclResult r = a->AllReduce(buf, buf, (size_t)count, kRcclFloat32, kRcclSum, x->comm, st);
  if (r != kRcclSuccess) return fail("ncclAllReduce", r);
  return 0;
}
int cb_wait_all(void* c, void* stream) {
  Ctx* x = (Ctx*)c;
  if (!x) return 1;
  int rc = 0;
  for (auto& p : x->pending) {
    if (p.first != (hipStream_t)stream &&            // the same stream is already ordered
        (hipEventRecord(p.second, p.first) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, p.second, 0) != hipSuccess)) {
      g_rccl_err = "hipEventRecord / hipStreamWaitEvent failed in wait_all";
      (void)hipGetLastError();
      rc = 1;
    }
    x->free_events.push_back(p.second);
  }
  x->pending.clear();
  return rc;
}
int cb_allgather(void* c, const float* src, float* dst, int64_t count, void* stream) {
  Ctx* x = (Ctx*)c;
  Api* a = api();
  if (!x || !a || !src || !dst || count <= 0) return 1;
  const RcclResult r = a->AllGather(src, dst, (size_t)count, kRcclFloat32, x->gather ? x->gather : x->comm, (hipStream_t)stream);
  if (r != kRcclSuccess) return fail("ncclAllGather", r);
  return 0;
}

}  // namespace

extern "C" {

const char* ian_rccl_last_error(void) { return g_rccl_err.c_str(); }

/* every rank, before anything else: can librccl be loaded at all?  Touches no communicator and never blocks, so the ranks can agree on
   the outcome before any of them enters ncclCommInitRank (which blocks until ALL ranks have arrived).  0 / -10. */
int ian_rccl_available(void) {
  if (api()) return 0;
  g_rccl_err = "librccl.so could not be loaded (dlopen) or lacks an entry point";
  return -10;
}

/* rank 0: 128 bytes to hand to every rank (ncclGetUniqueId).  -10: librccl could not be loaded. */
int ian_rccl_unique_id(void* out128) {
  Api* a = api();
  if (!out128) return -1;
  if (!a) {
    g_rccl_err = "librccl.so could not be loaded (dlopen)";
    return -10;
  }
  RcclUniqueId id;
  const RcclResult r = a->GetUniqueId(&id);
  if (r != kRcclSuccess) return -fail("ncclGetUniqueId", r);
  memcpy(out128, id.internal, kIdBytes);
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
  RcclUniqueId id;
  memcpy(id.internal, id128, kIdBytes);
  Ctx* x = new Ctx();
  x->world = world;
  x->rank = rank;
  const RcclResult r = a->CommInitRank(&x->comm, world, id, rank);
  if (r != kRcclSuccess) {
    delete x;
    return -fail("ncclCommInitRank", r);
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

/* optional, every rank, after ian_rccl_comm_create: a second communicator (its own id from ian_rccl_unique_id) that carries the
   all-gathers, so that they are not queued behind gradient buckets in flight on the first one.  Blocks like comm_create. */
int ian_rccl_comm_add_gather(ian_comm_ops* ops, const void* id128) {
  Api* a = api();
  if (!ops || !ops->ctx || !id128 || ops->allgather != cb_allgather) return -1;
  if (!a) return -10;
  Ctx* x = (Ctx*)ops->ctx;
  if (x->gather) return -6;
  RcclUniqueId id;
  memcpy(id.internal, id128, kIdBytes);
  const RcclResult r = a->CommInitRank(&x->gather, x->world, id, x->rank);
  if (r != kRcclSuccess) {
    x->gather = nullptr;
    return -fail("ncclCommInitRank (gather communicator)", r);
  }
  return 0;
}

/* after ian_trainer_destroy of every trainer that used the table */
void ian_rccl_comm_destroy(ian_comm_ops* ops) {
  if (!ops || !ops->ctx) return;
  Ctx* x = (Ctx*)ops->ctx;
  Api* a = api();
  for (auto& p : x->pending) (void)hipEventDestroy(p.second);
  for (hipEvent_t e : x->free_events) (void)hipEventDestroy(e);
  if (a && x->gather) (void)a->CommDestroy(x->gather);
  if (a && x->comm) (void)a->CommDestroy(x->comm);
  delete x;
  ops->ctx = nullptr;
}

}  // extern "C"
