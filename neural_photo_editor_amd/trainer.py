"""Training step of the full IAN on MI355X: the host-side caller of ``ian_train_step`` (include/ian_train.h,
csrc/ian_trainer.cpp), the equivalent of train_IAN.py:47-352 (``make_training_functions`` -> ``update_gen`` /
``update_discrim``).

Round 4: there is ONE sequencer of the step, the C++ one.  The reference builds one Theano graph out of Lasagne layers and
lets Theano differentiate it; csrc/ian_trainer.cpp wires the same graph explicitly, forward and backward, over the
``ian_layer_*`` / ``ian_k_*`` entry points, including the data-parallel machinery (gradient buckets handed to the all-reduce
while backward still runs, SyncBN statistics combined in rank order, MinibatchLayer all-gather).  This module

* loads the parameters (Theano names, GANcheckpoints layout) and the MADE masks into an ``ian_trainer``;
* fills the trainer's collective callback table (``ian_comm_ops``) from ``torch.distributed`` -- backend "nccl" is RCCL
  over xGMI, gloo in the tests -- so that the SAME C entry runs at world size 1 and N (``Comm.ops``);
* exposes the step (``step`` / ``update_gen`` / ``update_discrim``), its pieces (``forward``, ``metrics``, ``backward``,
  ``_finish_allreduce``, ``_regularizers``) and zero-copy views of the trainer's device buffers for the parity tests;
* keeps the thin ctypes wrappers of the building blocks (``Layer``, ``K``, ``BN``) the kernel-level tests drive directly.

Python sequences nothing and does no arithmetic on activations, gradients or parameters.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import config_loader, made
from .lib import OpDesc, load_train_library

BN_EPS = 1e-4
ACT = {"none": 0, "relu": 1, "lrelu": 2, "elu": 3, "tanh": 4, "sigmoid": 5}
K_CONV, K_DECONV, K_MDC, K_DENSE = 1, 2, 3, 4

ENC_WIDTHS = (128, 256, 512, 1024)
DEC_STAGES = (("dec_conv1", 512, 512, 4, "dec_conv2a", [0, 2]), ("dec_conv2", 512, 256, 8, "dec_conv3a", [0, 2, 3]),
              ("dec_conv3", 256, 128, 16, "dec_conv4a", [0, 2, 3]))   # (deconv, cin, cout, in_hw, block, scales) IAN.py:139-171
HEAD_SCALES = [2, 3, 4]


def cs(c):
    return (c + 31) // 32 * 32


def mdcl_names(name, scales):
    return [name + "W", name + "_coeff_base"] + [name + ("_coeff_1x1" if s == 0 else "_coeff_%d" % s) for s in scales]


# ======================================================================================================
# communication (RCCL through torch.distributed; gloo in the CPU tests)
# ======================================================================================================
class Comm:
    """The collective table of the C trainer (include/ian_train.h: ian_comm_ops) filled from ``torch.distributed`` -- backend
    "nccl" is RCCL over xGMI; gloo in the tests.  ``bucket_bytes`` sizes the gradient buckets the C sequencer cuts for xGMI
    (point-to-point links: a few large messages).

    The batch-statistics / MinibatchLayer all-gathers of ``exact`` mode get their OWN process group (``gather_group``):
    torch's NCCL backend runs every collective of one group on one internal stream in issue order, so on a shared group a
    16 MB gradient bucket handed over during backward would sit in front of the next small all-gather the compute stream
    blocks on.  ``filler`` names what ended up carrying the collectives (bench.py prints it)."""

    filler = "torch.distributed"

    def __init__(self, group=None, bucket_bytes=16 << 20, gather_group="own"):
        import torch.distributed as dist
        self.dist = dist
        self.active = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.bucket_bytes = bucket_bytes
        if gather_group == "own":              # collective call: every rank constructs its Comm
            gather_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None) \
                if self.active and self.world > 1 else group
        self.gather_group = gather_group
        self.errors = []

    # ---- the collective callback table of the C trainer ----------------------------------------------------------------
    def ops(self, torch, host=False):
        """ian_comm_ops filled from torch.distributed: the trainer calls these with raw device pointers and HIP streams
        (host=True: raw HOST pointers, streams ignored -- what the CPU gloo tests drive the table with).
        allreduce_sum: async all-reduce issued with the trainer's side stream current (RCCL orders it behind that stream and
        runs it on its own; gloo copies through the host) -- wait_all makes the compute stream wait for the works."""
        import contextlib
        self._works, self._views = [], {}

        def view(ptr, count):
            key = (int(ptr), int(count))
            t = self._views.get(key)
            if t is None:
                t = self._views[key] = host_view(torch, ptr, int(count)) if host else device_view(torch, ptr, (int(count),))
            return t

        def on(stream):
            if host:
                return contextlib.nullcontext()
            return torch.cuda.stream(torch.cuda.default_stream() if not stream else torch.cuda.ExternalStream(int(stream)))

        def allreduce(buf, count, stream):
            with on(stream):
                self._works.append(self.dist.all_reduce(view(buf, count), op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

        def wait_all(stream):
            with on(stream):
                for w in self._works:
                    w.wait()                      # RCCL: the given stream waits (device side); gloo: host wait
            self._works = []

        def allgather(src, dst, count, stream):
            with on(stream):
                self.all_gather_rows(view(src, count).view(1, -1), view(dst, count * self.world).view(self.world, -1))

        self.errors = []
        return build_ops(self.world, self.rank, allreduce, wait_all, allgather, self.errors)

    def all_gather_rows(self, local, out):
        """out[(rank*n):(rank+1)*n] = local over all ranks (row blocks of equal size), on the gather group."""
        if self.world == 1:
            out.copy_(local)
            return
        if self.dist.get_backend(self.gather_group) == "nccl":       # RCCL: one fused all-gather
            self.dist.all_gather_into_tensor(out, local.contiguous(), group=self.gather_group)
        else:                                                          # gloo (tests): list form, row blocks are views of `out`
            self.dist.all_gather(list(out.chunk(self.world, dim=0)), local.contiguous(), group=self.gather_group)

    def close(self):
        pass


class NativeRcclComm(Comm):
    """The same data-parallel step with the collective table filled by libian itself from librccl (csrc/ian_comm_rccl.cpp:
    ncclAllReduce on one communicator, ncclAllGather on a second one, on the trainer's streams) -- the route a C caller takes,
    and the default of bench.py / train_cli.py at N > 1.  torch.distributed is used ONLY to hand rank 0's two 128-byte
    communicator ids to the other ranks and to agree on the outcome of every set-up stage.

    ``one_comm=True`` (or IAN_RCCL_ONE_COMM=1): no second communicator -- the all-gathers share the gradient communicator, which then
    serialises everything in issue order.  Slower (a batch-statistics all-gather queues behind the buckets already handed over) but
    free of the co-residency assumption two communicators on one device make (include/ian_train.h, "CONTRACT"); bench.py falls back to
    it by itself when the two-communicator step does not come back.

    Set-up is STAGED so that no rank enters a blocking RCCL call unless every rank will: each stage is a local,
    non-blocking action followed by an agreement over torch.distributed (`_agree`), and only then the collective action all ranks
    enter together -- librccl loadable on every rank? -> ids from rank 0 -> ncclCommInitRank -> [second communicator] -> preflight
    buffers allocated? -> preflight collectives -> results right?  If any stage fails on any rank EVERY rank falls back to the
    torch.distributed filler of the base class together, and ``filler`` says why.  (A rank that dies INSIDE a blocking RCCL call
    cannot be agreed upon from here: that is what bench.py's per-attempt watchdog is for.)
    One GPU per rank (RCCL refuses two ranks on one device), so the shared-GPU gloo tests cannot drive the native table with two
    ranks: covered at world size 1 (tests/test_gpu_dp.py), by the staged-fallback tests (tests/test_comm.py), and by construction."""

    def __init__(self, group=None, bucket_bytes=16 << 20, one_comm=None):
        import os
        Comm.__init__(self, group, bucket_bytes, gather_group=None)   # a gather GROUP is created only if the fallback needs it
        self.one_comm = bool(int(os.environ.get("IAN_RCCL_ONE_COMM", "0"))) if one_comm is None else bool(one_comm)
        self.filler = "librccl (native, %s)" % ("1 communicator" if self.one_comm else "2 communicators")
        self._native_ops = None
        self.stages = []              # (stage, ok) as agreed by all ranks: what the tests and the bench line can look at

    def _agree(self, stage, ok, why=""):
        """All ranks learn whether `stage` worked everywhere.  -> (ok everywhere, first reason)."""
        if self.world > 1:
            oks = [None] * self.world
            self.dist.all_gather_object(oks, (bool(ok), str(why)), group=self.group)
            bad = ["rank %d: %s" % (r, w or "failed") for r, (k, w) in enumerate(oks) if not k]
            ok, why = (not bad), (bad[0] if bad else "")
        self.stages.append((stage, bool(ok)))
        return bool(ok), why

    def _test_fault(self, stage):
        """IAN_COMM_TEST_FAIL=<stage>[@rank]: the CPU tests inject a failure of one stage on one rank (or all)."""
        import os
        spec = os.environ.get("IAN_COMM_TEST_FAIL", "")
        if not spec:
            return False
        st, _, rk = spec.partition("@")
        return st == stage and (rk == "" or int(rk) == self.rank)

    def ops(self, torch, host=False):
        lib = load_train_library()
        err = lambda: (lib.ian_rccl_last_error() or b"?").decode()
        self.stages = []

        def fallback(why):
            self.close()
            if self.world == 1:
                raise IanTrainError(why or "librccl communicator could not be created")
            self.filler = "torch.distributed (fallback: %s)" % why
            # IAN_RCCL_ONE_COMM also applies to the fallback: no second process group either
            self.gather_group = self.group if self.one_comm else \
                self.dist.new_group(ranks=self.dist.get_process_group_ranks(self.group) if self.group is not None else None)
            return Comm.ops(self, torch, host=host)

        # stage 0 -- local, never blocks: can this rank use librccl at all?
        why = ""
        if host:
            why = "host buffers requested"
        elif self._test_fault("available") or lib.ian_rccl_available() != 0:
            why = "librccl not loadable: %s" % err()
        ok, why = self._agree("available", not why, why)
        if not ok:
            return fallback(why)
        # stage 1 -- rank 0 draws the ids (local), everybody receives them (collective on torch.distributed, all ranks enter: agreed above)
        ids, why = None, ""
        if self.rank == 0:
            bufs = [C.create_string_buffer(128) for _ in range(1 if self.one_comm else 2)]
            rcs = [lib.ian_rccl_unique_id(b) for b in bufs]
            if any(rcs) or self._test_fault("ids"):
                why = "ian_rccl_unique_id failed (%s): %s" % (rcs, err())
            else:
                ids = [b.raw for b in bufs]
        if self.world > 1:
            box = [ids]
            self.dist.broadcast_object_list(box, src=0, group=self.group)
            ids = box[0]
        ok, why = self._agree("ids", ids is not None, why)
        if not ok:
            return fallback(why)
        # stage 2 -- ncclCommInitRank: every rank enters (blocks until all have)
        o = CommOps()
        rc = -99 if self._test_fault("comm_create") else lib.ian_rccl_comm_create(C.create_string_buffer(ids[0], 128), self.rank, self.world, C.byref(o))
        if rc == 0:
            self._lib, self._native_ops = lib, o
        ok, why = self._agree("comm_create", rc == 0, "ian_rccl_comm_create failed (%d): %s" % (rc, err()) if rc else "")
        if not ok:
            return fallback(why)
        # stage 3 -- the all-gathers' own communicator (skipped in one-communicator mode)
        if not self.one_comm:
            rc = -99 if self._test_fault("add_gather") else lib.ian_rccl_comm_add_gather(C.byref(o), C.create_string_buffer(ids[1], 128))
            ok, why = self._agree("add_gather", rc == 0, "ian_rccl_comm_add_gather failed (%d): %s" % (rc, err()) if rc else "")
            if not ok:
                return fallback(why)
        # stage 4 -- preflight: buffers first (local; may fail), agreement, then the collectives all ranks enter, then the verdict
        pre, why = None, ""
        try:
            if self._test_fault("preflight_alloc"):
                raise RuntimeError("injected")
            pre = self._preflight_alloc(torch)
        except BaseException as exc:              # noqa: BLE001 -- every rank must reach the agreement below
            why = "preflight buffers: %r" % (exc,)
        ok, why = self._agree("preflight_alloc", pre is not None, why)
        if not ok:
            return fallback(why)
        why = self._preflight_run(torch, o, pre)
        ok, why = self._agree("preflight", not why, why)
        if not ok:
            return fallback(why)
        self.errors = []
        return o

    def _preflight_alloc(self, torch):
        w, r, n = self.world, self.rank, 1024
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            a = torch.full((n,), float(r + 1), device="cuda")
        with torch.cuda.stream(s2):
            src = torch.full((n,), float(r + 1), device="cuda")
            dst = torch.zeros(w * n, device="cuda")
        torch.cuda.synchronize()
        return (s1, s2, a, src, dst, n)

    def _preflight_run(self, torch, o, pre):
        """One all-reduce and one all-gather of known values through the freshly filled table, each on its own side stream, before
        the trainer sees it: the local NCCL ABI declarations of csrc/ian_comm_rccl.cpp (datatype / reduction codes, argument
        order) are only exercised with more than one rank here, and a wrong sum must send every rank to the torch.distributed
        filler instead of into the gradients.  Returns "" or the reason."""
        try:
            s1, s2, a, src, dst, n = pre
            w = self.world
            rcs = (o.allreduce_sum(o.ctx, a.data_ptr(), n, s1.cuda_stream),
                   o.allgather(o.ctx, src.data_ptr(), dst.data_ptr(), n, s2.cuda_stream),
                   o.wait_all(o.ctx, s2.cuda_stream))
            torch.cuda.synchronize()
            if any(rcs):
                return "preflight collective failed %s: %s" % (rcs, (self._lib.ian_rccl_last_error() or b"?").decode())
            want = torch.arange(1, w + 1, device="cuda", dtype=torch.float32).repeat_interleave(n)
            if not bool((a == w * (w + 1) / 2).all()):
                return "preflight all-reduce returned %r, expected %r" % (float(a[0]), w * (w + 1) / 2)
            if not torch.equal(dst, want):
                return "preflight all-gather returned wrong rows"
            return ""
        except BaseException as exc:              # noqa: BLE001 -- every rank must reach the agreement
            return "preflight raised %r" % (exc,)

    def close(self):
        if getattr(self, "_native_ops", None) is not None:
            self._lib.ian_rccl_comm_destroy(C.byref(self._native_ops))
            self._native_ops = None


COMM_MODES = ("native2", "native1", "torch")   # bench.py's retry ladder walks them in this order


def default_comm(bucket_bytes=16 << 20, mode=None):
    """The communicator bench.py / train_cli.py use.  mode (argument, else IAN_COMM_MODE, else "native2"):
        native2  the torch-free RCCL filler with the all-gathers on a second communicator (NativeRcclComm);
        native1  the same with ONE communicator (also selected by IAN_RCCL_ONE_COMM=1);
        torch    the torch.distributed filler (one process group per collective kind, or one in all with IAN_RCCL_ONE_COMM=1).
    The native fillers need torch.distributed on RCCL ("nccl") and more than one rank; otherwise (gloo: CPU rehearsals, ranks
    sharing one GPU; no process group) the torch.distributed filler is returned whatever the mode."""
    import os
    import torch.distributed as dist
    mode = mode or os.environ.get("IAN_COMM_MODE") or "native2"
    if mode not in COMM_MODES:
        raise ValueError("comm mode %r: expected one of %s" % (mode, ", ".join(COMM_MODES)))
    one = bool(int(os.environ.get("IAN_RCCL_ONE_COMM", "0")))
    if mode != "torch" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl":
        return NativeRcclComm(bucket_bytes=bucket_bytes, one_comm=(mode == "native1") or one)
    c = Comm(bucket_bytes=bucket_bytes, gather_group=None if one else "own")
    if one:
        c.filler = "torch.distributed (one process group)"
    return c


# ======================================================================================================
# thin wrappers over the C ABI
# ======================================================================================================
class IanTrainError(RuntimeError):
    pass


def _p(t):
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


class Layer:
    """ian_layer (include/ian_train.h): one linear Lasagne layer with packed weights on the device."""

    def __init__(self, lib, kind, cin, cout, in_h=1, in_w=1, scales=(), flat=(0, 0, 0), unflat=(0, 0, 0), deconv_flip=True):
        self.lib = lib
        d = OpDesc()
        d.kind, d.cin, d.cout, d.in_h, d.in_w = kind, cin, cout, in_h, in_w
        d.src = d.dst = 0
        d.src2 = d.src3 = -1
        d.flat_c, d.flat_h, d.flat_w = flat
        d.unflat_c, d.unflat_h, d.unflat_w = unflat
        d.n_scales = len(scales)
        for i, s in enumerate(scales):
            d.scales[i] = s
        self.h = C.c_void_p()
        rc = lib.ian_layer_create(C.byref(d), int(bool(deconv_flip)), C.byref(self.h))
        if rc != 0:
            raise IanTrainError("ian_layer_create failed (%d)%s" % (rc, ": no HIP device, libian has no CPU fallback" if rc == -10 else ""))
        self.kind, self.cin, self.cout = kind, cin, cout
        self.nparams = lib.ian_layer_num_params(self.h)

    def _chk(self, rc):
        if rc != 0:
            raise IanTrainError("libian layer error %d: %s" % (rc, (self.lib.ian_layer_last_error(self.h) or b"?").decode()))

    def _ptrs(self, tensors):
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        return arr

    def set_params(self, tensors, stream=0):
        self._chk(self.lib.ian_layer_set_params(self.h, self._ptrs(tensors), len(tensors), C.c_void_p(stream)))

    def forward(self, x, n, y, y_stride=0, bias=None, res=None, act=0, stream=0):
        self._chk(self.lib.ian_layer_forward(self.h, _p(x), n, _p(y), y_stride, _p(bias), _p(res), act, C.c_void_p(stream)))

    def backward_data(self, dy, n, dx, dx_stride=0, accumulate=False, stream=0):
        self._chk(self.lib.ian_layer_backward_data(self.h, _p(dy), n, _p(dx), dx_stride, int(accumulate), C.c_void_p(stream)))

    def backward_weight(self, x, dy, n, dparams, accumulate=False, stream=0):
        self._chk(self.lib.ian_layer_backward_weight(self.h, _p(x), _p(dy), n, self._ptrs(dparams), len(dparams), int(accumulate),
                                                     C.c_void_p(stream)))

    def head6_forward(self, l1, l2, x, n, y0, y1, y2, y_stride, acts, stream=0):
        """Three sibling 2-filter MDCL layers in one pass (ian_layer_head6_forward); False when the shape does not qualify."""
        rc = self.lib.ian_layer_head6_forward(self.h, l1.h, l2.h, _p(x), n, _p(y0), _p(y1), _p(y2), y_stride, acts[0], acts[1], acts[2],
                                              C.c_void_p(stream))
        if rc == -4:
            return False
        self._chk(rc)
        return True

    def head6_backward(self, l1, l2, x, dys, n, dy_stride, dx=None, dx_stride=0, dx_accumulate=False, dparams3=None, accumulate=False,
                       stream=0):
        """Backward of the same three layers through one shifted gather + dense GEMMs (ian_layer_head6_backward): data gradient
        into dx and/or weight gradients into dparams3 = [[5 tensors] x 3]; False when the shape does not qualify."""
        pp = [self._ptrs(d) for d in dparams3] if dparams3 else [None, None, None]
        rc = self.lib.ian_layer_head6_backward(self.h, l1.h, l2.h, _p(x), _p(dys[0]), _p(dys[1]), _p(dys[2]), n, dy_stride, _p(dx),
                                               dx_stride, int(dx_accumulate), pp[0], pp[1], pp[2], len(dparams3[0]) if dparams3 else 0,
                                               int(accumulate), C.c_void_p(stream))
        if rc == -4:
            return False
        self._chk(rc)
        return True

    def autotune(self, n, scratch_a, scratch_b, stream=0):
        self._chk(self.lib.ian_layer_autotune(self.h, n, _p(scratch_a), _p(scratch_b), min(scratch_a.numel(), scratch_b.numel()),
                                              C.c_void_p(stream)))

    def close(self):
        if self.h:
            self.lib.ian_layer_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class K:
    """ian_k_* launches with tensor arguments."""

    def __init__(self, lib, stream=0):
        self.lib = lib
        self.stream = C.c_void_p(stream)

    def __getattr__(self, name):
        fn = getattr(self.lib, "ian_k_" + name)

        def call(*args):
            conv = []
            for a, t in zip(args, fn.argtypes):
                if t is C.c_void_p:
                    conv.append(_p(a))
                else:
                    conv.append(a)
            rc = fn(*conv, self.stream)
            if rc != 0:
                raise IanTrainError("ian_k_%s failed (%d): %s" % (name, rc, (self.lib.ian_k_last_error() or b"?").decode()))
        return call


class BN:
    """Lasagne batch_norm in training mode (App. B.3): state of one normalisation in one pass."""

    def __init__(self, torch, C, device):
        self.C = C
        z = lambda: torch.zeros(C, dtype=torch.float32, device=device)
        # float64 column sums (kernels_train.hip NUMERICS: every element is widened before it is squared / added)
        self.sums, self.bsums = torch.zeros(2 * C, dtype=torch.float64, device=device), torch.zeros(2 * C, dtype=torch.float64, device=device)
        self.mean, self.inv_std, self.scale, self.shift = z(), z(), z(), z()
        self.count = 1.0


# ======================================================================================================
# zero-copy torch views of device memory owned by libian (gradient buckets for torch.distributed, test introspection)
# ======================================================================================================
class _CAI:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType), ("shape", C.POINTER(C.c_int64)),
                ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DLDeleter = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DLDeleter)]
_dl_keep = {}


@_DLDeleter
def _dl_delete(mt):                                   # torch calls this when the aliasing tensor dies: drop our bookkeeping only
    _dl_keep.pop(C.addressof(mt.contents), None)


def _view_dlpack(torch, ptr, n, dtype):
    """DLPack capsule over libian-owned device memory (kDLROCM), for torch builds whose __cuda_array_interface__ import is
    unavailable; the memory is NOT freed by the deleter."""
    shape = (C.c_int64 * 1)(n)
    mt = _DLManagedTensor()
    mt.dl_tensor.data = C.c_void_p(int(ptr))
    mt.dl_tensor.device = _DLDevice(10, torch.cuda.current_device())          # kDLROCM
    mt.dl_tensor.ndim = 1
    mt.dl_tensor.dtype = _DLDataType(2, 64 if dtype == "f8" else 32, 1)       # kDLFloat
    mt.dl_tensor.shape = C.cast(shape, C.POINTER(C.c_int64))
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _dl_delete
    _dl_keep[C.addressof(mt)] = (mt, shape)
    new = C.pythonapi.PyCapsule_New
    new.restype, new.argtypes = C.py_object, [C.c_void_p, C.c_char_p, C.c_void_p]
    return torch.from_dlpack(new(C.addressof(mt), b"dltensor", None))


_view_mode = [None]


def device_view(torch, ptr, shape, dtype="f4"):
    """torch tensor over ``ptr`` (device memory that outlives the view; libian owns it) -- no copy."""
    n = int(np.prod(shape)) if len(shape) else 1
    if n == 0 or not ptr:
        raise IanTrainError("device_view: null pointer or empty shape")
    t = None
    if _view_mode[0] in (None, "cai"):
        try:
            t = torch.as_tensor(_CAI(ptr, (n,), "<" + dtype), device="cuda")
            if t.data_ptr() != int(ptr):
                t = None
            else:
                _view_mode[0] = "cai"
        except Exception:
            t = None
    if t is None:
        t = _view_dlpack(torch, ptr, n, dtype)
        if t.data_ptr() != int(ptr):
            raise IanTrainError("device_view: torch copied the buffer instead of aliasing it")
        _view_mode[0] = "dlpack"
    return t.view(*shape)


def host_view(torch, ptr, n, dtype=np.float32):
    """torch tensor over n elements of HOST memory at ``ptr`` (no copy; the memory must outlive the view)."""
    if not ptr or n <= 0:
        raise IanTrainError("host_view: null pointer or empty shape")
    ct = {np.float32: C.c_float, np.float64: C.c_double}[dtype]
    return torch.from_numpy(np.ctypeslib.as_array(C.cast(C.c_void_p(int(ptr)), C.POINTER(ct)), shape=(int(n),)))


class CommOps(C.Structure):
    """ian_comm_ops (include/ian_train.h)"""
    AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    WA = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("ctx", C.c_void_p), ("allreduce_sum", AR), ("wait_all", WA), ("allgather", AG)]


def build_ops(world, rank, allreduce, wait_all, allgather, errors):
    """ian_comm_ops from three Python callables: allreduce(ptr, count, stream), wait_all(stream), allgather(src, dst, count, stream)
    (raw device addresses / HIP stream handles as ints).  Exceptions never cross the C boundary: they are appended to ``errors``
    and reported to the trainer as a non-zero return code (the step then fails with -30)."""
    def guard(fn):
        def wrapped(ctx, *a):
            try:
                fn(*a)
                return 0
            except BaseException as exc:          # noqa: BLE001 -- must not propagate into C
                errors.append(exc)
                return 1
        return wrapped

    o = CommOps()
    o.world, o.rank, o.ctx = int(world), int(rank), None
    o.allreduce_sum, o.wait_all, o.allgather = CommOps.AR(guard(allreduce)), CommOps.WA(guard(wait_all)), CommOps.AG(guard(allgather))
    o._keep = (o.allreduce_sum, o.wait_all, o.allgather)   # ctypes does not keep callbacks stored in a struct alive by itself
    return o


class TrainConfig(C.Structure):
    _fields_ = [("batch", C.c_int32), ("num_latents", C.c_int32), ("deconv_flip", C.c_int32), ("reserved", C.c_int32),
                ("learning_rate", C.c_double), ("beta1", C.c_double), ("reg", C.c_float), ("ortho", C.c_float), ("recon_weight", C.c_float),
                ("feature_weight", C.c_float), ("dg_weight", C.c_float), ("dd_weight", C.c_float), ("agr_weight", C.c_float),
                ("ags_weight", C.c_float)]


METRICS = ("discrim_d_loss", "gen_recon_loss", "gen_sample_loss", "discrim_g_loss", "discrim_acc", "kl_div", "pixel_loss",
           "pixel_acc", "feature_loss")
GEN_KEYS = ("gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc")          # train_IAN.py:291-296
DISCRIM_KEYS = ("discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc")        # train_IAN.py:298-304
GROUP_INDEX = {"enc": 0, "Z": 1, "dec": 2}


class _Group:
    """One Adam group of the C trainer: names, offsets and zero-copy views of its flat device buffers."""

    def __init__(self, tr, gname):
        self.tr, self.gname, self.index = tr, gname, GROUP_INDEX[gname]
        self.names, self.offsets = [], {}
        self._views = None

    @property
    def numel(self):
        self._load()
        return self._numel

    def _load(self):
        if self._views is None:
            ptrs = [C.c_void_p() for _ in range(4)]
            n = C.c_int64()
            self.tr._check(self.tr.lib.ian_trainer_group(self.tr._h, self.index, *[C.byref(q) for q in ptrs], C.byref(n)))
            self._numel = n.value
            self._views = [device_view(self.tr.torch, q.value, (n.value,)) for q in ptrs]
        return self._views

    p = property(lambda self: self._load()[0])
    g = property(lambda self: self._load()[1])
    m = property(lambda self: self._load()[2])
    v = property(lambda self: self._load()[3])
    t = property(lambda self: int(self.tr.lib.ian_trainer_adam_steps(self.tr._h, self.index)))

    def view(self, buf, name):
        o, cnt, shape = self.offsets[name]
        return buf[o:o + cnt]


class _Pass:
    """tr.EX['a4'], tr.DZ['xhat'] ...: shaped zero-copy views of the buffers of one network pass inside the C trainer."""

    def __init__(self, tr, tag):
        self.tr, self.tag, self._cache = tr, tag, {}

    def _shape(self, name, numel):
        n = self.tr.n
        if self.tag[0] == "E":
            if name in ("x", "dx"):
                return (n, 64, 64, 32)
            if name[-1] in "1234" and name[:-1] in ("a", "da", "y"):
                i = int(name[-1]) - 1
                return (n, 32 >> i, 32 >> i, ENC_WIDTHS[i])
        if self.tag[0] == "D":
            if name in ("xhat", "dxhat", "tmp_img"):
                return (n, 3, 64, 64)
            if name in ("h0", "dh0"):
                return (n, 4, 4, 512)
            if name in ("y4", "h4", "dh4"):
                return (n, 64, 64, 128)
            for dc, ci, co, hw, blk, sc in DEC_STAGES:
                if name.startswith(blk + "_"):
                    return (n, 2 * hw, 2 * hw, co)
            if numel == n * 4096 * 32:
                return (n, 64, 64, 32)
        if numel % n == 0:
            return (n, numel // n)
        return (numel,)

    def __getitem__(self, name):
        if name not in self._cache:
            ptr, cnt = C.c_void_p(), C.c_int64()
            self.tr._check(self.tr.lib.ian_trainer_buffer(self.tr._h, ("%s.%s" % (self.tag, name)).encode(), C.byref(ptr), C.byref(cnt)))
            f64 = name.endswith((".sums", ".bsums"))
            self._cache[name] = device_view(self.tr.torch, ptr.value, (cnt.value,) if f64 else self._shape(name, cnt.value), "f8" if f64 else "f4")
        return self._cache[name]


class Trainer:
    """update_gen / update_discrim of train_IAN.py on one GPU, or on one rank of a data-parallel job: a thin caller of the
    C++ sequencer (csrc/ian_trainer.cpp).  ``batch`` is the per-rank batch; ``comm`` a ``Comm`` (default: the current
    torch.distributed state, world 1 when it is not initialised)."""

    def __init__(self, config_path, params, batch, comm=None, exact=True, device="cuda", deconv_flip=True):
        import torch
        self.torch = torch
        self.lib = load_train_library()
        self.dev = torch.device(device)
        c = dict(config_loader.load_config(config_path).cfg)
        self.cfg = c
        if int(c["num_latents"]) != 100:
            raise IanTrainError("the training graph is IAN.py's: num_latents must be 100, got %r" % (c["num_latents"],))
        lr = c["learning_rate"][0] if isinstance(c["learning_rate"], dict) else c["learning_rate"]
        self.n = int(batch)
        self.comm = comm or Comm()
        self.exact = bool(exact) and self.comm.world > 1
        self.N = self.n * self.comm.world
        self.zdim = int(c["num_latents"])
        tc = TrainConfig(self.n, self.zdim, int(bool(deconv_flip)), 0, float(lr), float(c["beta1"]), float(c["reg"]),
                         float(c.get("ortho", -1.0)), float(c["recon_weight"]), float(c["feature_weight"]), float(c["dg_weight"]),
                         float(c["dd_weight"]), float(c["agr_weight"]), float(c["ags_weight"]))
        self._lr = float(lr)
        self._h = C.c_void_p()
        rc = self.lib.ian_trainer_create(C.byref(tc), C.byref(self._h))
        if rc:
            raise IanTrainError("ian_trainer_create failed (%d)%s" % (rc, ": no HIP device, libian has no CPU fallback" if rc == -10 else ""))
        self._ops = None
        if self.comm.world > 1:
            self._ops = self.comm.ops(torch)                      # keeps the callbacks alive as long as the trainer
            self._check(self.lib.ian_trainer_set_comm(self._h, C.byref(self._ops), int(self.exact)))
        self.shapes = {}
        self.frozen = {k: np.asarray(v, np.float32) for k, v in params.items() if k.startswith("l_IAF_")}
        for name, v in params.items():
            a = np.ascontiguousarray(v, np.float32)
            rc = self.lib.ian_trainer_load_param(self._h, name.encode(), C.c_void_p(a.ctypes.data), a.size)
            if rc == -2:
                continue                                          # an entry the training graph does not own
            self._check(rc)
            self.shapes[name] = a.shape
        m = [np.ascontiguousarray(x, np.float32) for x in made.masks_once(self.zdim)]      # train_IAN.py:404-405
        self._check(self.lib.ian_trainer_set_made_masks(self._h, *[C.c_void_p(x.ctypes.data) for x in m], m[0].shape[0]))
        self._check(self.lib.ian_trainer_finalize(self._h))
        if self.comm.world > 1 and getattr(self.comm, "bucket_bytes", None):
            self.set_option("bucket_bytes", int(self.comm.bucket_bytes))       # Comm(bucket_bytes=...) sizes the C sequencer's buckets
        self.groups = {g: _Group(self, g) for g in ("enc", "Z", "dec")}
        self.where = {}
        gi, off, cnt = C.c_int32(), C.c_int64(), C.c_int64()
        for name, shp in self.shapes.items():
            if name.startswith("l_IAF_"):
                continue
            self._check(self.lib.ian_trainer_param_info(self._h, name.encode(), C.byref(gi), C.byref(off), C.byref(cnt)))
            if gi.value < 3:
                g = self.groups[("enc", "Z", "dec")[gi.value]]
                g.names.append(name)
                g.offsets[name] = (off.value, cnt.value, tuple(shp))
                self.where[name] = g.gname
        for g in self.groups.values():                            # Lasagne's topological order = ascending offsets
            g.names.sort(key=lambda nme: g.offsets[nme][0])
        self.EX, self.EH, self.EG = _Pass(self, "EX"), _Pass(self, "EH"), _Pass(self, "EG")
        self.ZS, self.DZ, self.DG = _Pass(self, "ZS"), _Pass(self, "DZ"), _Pass(self, "DG")
        self.k = K(self.lib, 0)
        self._keep = None

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc:
            msg = (self.lib.ian_trainer_last_error(self._h) or b"?").decode()
            errs = getattr(self.comm, "errors", None)
            if rc == -30 and errs:
                msg += " <- %s: %s" % (type(errs[-1]).__name__, errs[-1])
            raise IanTrainError("libian trainer error %d: %s" % (rc, msg))

    def _arg(self, a, shape, what):
        """float32 C-contiguous (n, ...) numpy array or torch tensor (host or device) -> (keep-alive object, pointer)"""
        if hasattr(a, "data_ptr"):
            if a.dtype != self.torch.float32 or not a.is_contiguous() or tuple(a.shape) != shape:
                raise IanTrainError("%s must be a contiguous float32 tensor of shape %s, got %s %s%s" % (
                    what, shape, a.dtype, tuple(a.shape), "" if a.is_contiguous() else " (non-contiguous)"))
            return a, C.c_void_p(a.data_ptr())
        arr = np.ascontiguousarray(a, np.float32)
        if arr.shape != shape:
            raise IanTrainError("%s must have shape %s, got %s" % (what, shape, arr.shape))
        return arr, C.c_void_p(arr.ctypes.data)

    def _inputs(self, X, Z, eps):
        n = self.n
        if int(np.shape(X)[0]) != n:
            raise IanTrainError("batch %d, the trainer was created for %d per GPU" % (int(np.shape(X)[0]), n))
        keep = [self._arg(X, (n, 3, 64, 64), "X"), self._arg(Z, (n, self.zdim), "Z"), self._arg(eps, (n, self.zdim), "eps")]
        self._keep = keep                                         # the step is asynchronous: keep host / device buffers alive
        return [p for _, p in keep]

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):                                          # train_IAN.py:523-527 learning-rate schedule
        self._lr = float(value)
        self.set_option("learning_rate", float(value))

    def set_option(self, key, value):
        self._check(self.lib.ian_trainer_set_option(self._h, key.encode(), float(value)))

    head6 = property(lambda self: True, lambda self, v: self.set_option("head6", int(bool(v))))
    overlap = property(lambda self: True, lambda self, v: self.set_option("overlap", int(bool(v))))
    update_running = property(lambda self: True, lambda self, v: self.set_option("update_running", int(bool(v))))
    measure_exposed = property(lambda self: False, lambda self, v: self.set_option("measure_exposed", int(bool(v))))

    def _named(self, name, shape=None):
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self.lib.ian_trainer_buffer(self._h, name.encode(), C.byref(ptr), C.byref(cnt)))
        return device_view(self.torch, ptr.value, shape or (cnt.value,))

    scalars = property(lambda self: self._named("scalars"))      # 64 floats of loss partials (tests)
    ws_loss = property(lambda self: self._named("ws_loss"))

    def stat(self, key):
        out = C.c_double()
        self._check(self.lib.ian_trainer_stat(self._h, key.encode(), C.byref(out)))
        return out.value

    # ---- the step ---------------------------------------------------------------------------------------------
    def step(self, which, X, Z, eps, return_metrics=True, stream=0):
        """which: 'gen' | 'discrim'.  X (n,3,64,64), Z (n,100), eps (n,100): float32 numpy arrays or tensors (host or device)."""
        px, pz, pe = self._inputs(X, Z, eps)
        out = (C.c_float * 9)() if return_metrics else None
        self._check(self.lib.ian_train_step(self._h, 0 if which == "gen" else 1, px, pz, pe, self.n, out, C.c_void_p(stream)))
        return dict(zip(METRICS, [float(v) for v in out])) if return_metrics else None

    def update_gen(self, X, Z, eps):
        """train_IAN.py:309-318 -> [gen_recon_loss, gen_sample_loss, pixel_loss, feature_loss, pixel_acc]"""
        m = self.step("gen", X, Z, eps)
        return [m[k] for k in GEN_KEYS]

    def update_discrim(self, X, Z, eps):
        """train_IAN.py:320-329 -> [discrim_g_loss, discrim_d_loss, discrim_acc, pixel_loss, pixel_acc]"""
        m = self.step("discrim", X, Z, eps)
        return [m[k] for k in DISCRIM_KEYS]

    def autotune(self, stream=0):
        """Pick tile shape / split-K / K-loop schedule per layer for this per-rank batch by timing the real launches on this
        GPU (untimed set-up work; summation order aside, results do not change)."""
        self._check(self.lib.ian_trainer_autotune(self._h, C.c_void_p(stream)))

    # ---- the step in pieces (parity tests) -----------------------------------------------------------------------
    def forward(self, X, Z, eps, xhat_override=None, xgen_override=None, stream=0):
        px, pz, pe = self._inputs(X, Z, eps)
        ov = []
        for o, what in ((xhat_override, "xhat_override"), (xgen_override, "xgen_override")):
            if o is None:
                ov.append(C.c_void_p(0))
            else:
                if not (hasattr(o, "is_cuda") and o.is_cuda):
                    raise IanTrainError("%s must be a device tensor" % what)
                keep, ptr = self._arg(o, (self.n, 3, 64, 64), what)
                self._keep.append((keep, ptr))
                ov.append(ptr)
        self._check(self.lib.ian_trainer_forward(self._h, px, pz, pe, self.n, ov[0], ov[1], C.c_void_p(stream)))

    def metrics(self):
        out = (C.c_float * 9)()
        self._check(self.lib.ian_trainer_metrics(self._h, out))
        return dict(zip(METRICS, [float(v) for v in out]))

    def backward(self, which):
        self._check(self.lib.ian_trainer_backward(self._h, 0 if which == "gen" else 1))

    def _finish_allreduce(self, which):
        self._check(self.lib.ian_trainer_finish_allreduce(self._h, 0 if which == "gen" else 1))

    def _regularizers(self, which):
        self._check(self.lib.ian_trainer_regularizers(self._h, 0 if which == "gen" else 1))

    def _apply_adam(self, which):
        self._check(self.lib.ian_trainer_apply_adam(self._h, 0 if which == "gen" else 1))

    def enc_backward(self, E, ce, feature_seeded, want_w, want_dx, reset=False):
        t0, w0, t1, w1 = ce
        self._check(self.lib.ian_trainer_enc_backward(self._h, {"EX": 0, "EH": 1, "EG": 2}[E.tag], int(t0), float(w0), int(t1), float(w1),
                                                      int(bool(feature_seeded)), int(bool(want_w)), int(bool(want_dx)), int(bool(reset))))

    def mark_params_changed(self, *groups):
        """Call after writing parameter values behind the trainer's back (tests, checkpoint loading)."""
        for g in groups or ("enc", "Z", "dec"):
            self._check(self.lib.ian_trainer_mark_dirty(self._h, GROUP_INDEX[g]))

    @property
    def overlap_log(self):
        out, rec = [], (C.c_int64 * 6)()
        for i in range(int(self.stat("overlap_log"))):
            if self.lib.ian_trainer_overlap_log(self._h, i, rec):
                break
            out.append({"which": ("gen", "discrim")[rec[0]], "bucket": (("enc", "Z", "dec")[rec[1]], int(rec[2])), "bytes": int(rec[3]),
                        "issued_at_write": int(rec[4]), "writes_in_backward": int(rec[5])})
        return out

    def plan_size(self, which):
        return int(self.stat("plan_buckets_" + which))

    def allreduce_exposed_ms(self):
        """Mean stall of the compute stream at wait_all per update kind -- the part of the gradient all-reduce backward did not
        hide, the wait_all stall ONLY (set measure_exposed = True first); ``allgather_ms`` is the other half of the picture."""
        return {w: self.stat("exposed_ms_" + w) for w in ("gen", "discrim")}

    def allgather_ms(self):
        """Mean time per update the compute stream spends inside the exact-mode all-gathers (batch statistics, MinibatchLayer),
        and their number per update (measure_exposed = True)."""
        return {w: {"ms": self.stat("gather_ms_" + w), "calls": self.stat("gathers_" + w)} for w in ("gen", "discrim")}

    # ---- parameters / checkpoints (GANcheckpoints.py format, Theano parameter names: train_IAN.py:563-569) -------
    def read(self, name, grad=False):
        out = np.empty(self.shapes[name], np.float32)
        self._check(self.lib.ian_trainer_read_param(self._h, name.encode(), int(grad), C.c_void_p(out.ctypes.data), out.size))
        return out

    def adam_steps(self):
        return tuple(self.lib.ian_trainer_adam_steps(self._h, g) for g in (0, 1, 2))

    def grads_numpy(self, gname):
        return {n: self.read(n, grad=True) for n in self.groups[gname].names}

    def params_numpy(self):
        return {n: self.read(n) for g in self.groups.values() for n in g.names}

    def state_dict(self):
        """Every parameter the inference graph needs, reference layout and names: trainable groups, batch-norm running
        averages (updated from the real-data pass, alpha 0.1) and the frozen MADE parameters."""
        out = {n: self.read(n) for n in self.shapes if not n.startswith("l_IAF_")}
        out.update(self.frozen)
        return out

    def save_weights(self, fname, metadata=None):
        from . import checkpoints
        meta = {"learning_rate": self.lr}
        meta.update(metadata or {})
        checkpoints.save_weights(fname, self.state_dict(), meta)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ian_trainer_destroy(self._h)
            self._h = C.c_void_p()
            if getattr(self, "_ops", None) is not None:      # the communicators die after the trainer that used their table
                getattr(self.comm, "close", lambda: None)()
                self._ops = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
