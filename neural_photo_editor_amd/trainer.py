"""Training step of the full IAN on MI355X: the host-side equivalent of train_IAN.py:47-352
(``make_training_functions`` -> ``update_gen`` / ``update_discrim``).

The reference builds ONE Theano graph out of Lasagne layers and lets Theano differentiate it; here the same
graph is wired explicitly, forward and backward, over the C ABI of include/ian_train.h: ``ian_layer_*`` objects
for the conv / transposed-conv / MDCL / dense layers (forward, backward-data, backward-weight = the three cuDNN /
GEMM calls Theano would emit) and ``ian_k_*`` launches for everything element-wise.  Python only sequences
launches and owns the device buffers (torch tensors are containers; no torch arithmetic touches activations,
gradients or parameters).

Graph (train_IAN.py:116-149): encoder(X) -> z ~ N(mu, e^ls) -> IAF -> decoder -> X_hat; encoder(X_hat);
decoder(IAF(Z)) -> X_gen; encoder(X_gen); every pass in batch-statistics batch-norm mode.
Updates (train_IAN.py:253-276): three Adam groups -- encoder_params (discriminator step), decoder_params
(generator step), Z_params (both).

Data parallel: one process per GPU, the minibatch is sharded; gradients are summed with a bucketed
RCCL all-reduce (``Comm``); with ``exact=True`` the batch-norm statistics are all-reduced and the MinibatchLayer
activations all-gathered so that the N-GPU step is the same function of the global minibatch as the 1-GPU step.
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
    """Sum-all-reduce / all-gather for the data-parallel step.  ``bucket_bytes`` sizes the gradient buckets for
    xGMI (point-to-point links: a few large messages)."""

    def __init__(self, group=None, bucket_bytes=16 << 20):
        import torch.distributed as dist
        self.dist = dist
        self.active = dist.is_available() and dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.bucket_bytes = bucket_bytes

    def all_reduce_sum(self, t, async_op=False):
        if self.world == 1:
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_reduce_buckets(self, flat, async_op=False):
        """flat: 1-D tensor; reduces it in bucket_bytes pieces; returns the work handles (async) or []."""
        if self.world == 1:
            return []
        n = flat.numel()
        step = max(1, self.bucket_bytes // flat.element_size())
        works = []
        for o in range(0, n, step):
            w = self.dist.all_reduce(flat[o:min(n, o + step)], op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                works.append(w)
        return works

    def all_reduce_sum_ordered(self, t, k):
        """t <- sum over ranks of t, combined in RANK ORDER by the same pairwise tree the per-rank reduction uses
        (all-gather + ian_k_tree_sum): the result does not depend on the collective's internal algorithm, and with
        power-of-two per-rank chunk counts it equals the single-process reduction bit for bit (SyncBN).
        Only used for the tiny per-channel statistic vectors (<= 2 x 16384 floats)."""
        if self.world == 1:
            return
        need = t.numel() * self.world
        buf = getattr(self, "_gbuf", None)
        if buf is None or buf.numel() < need or buf.device != t.device:
            buf = self._gbuf = t.new_empty(need)
        self.all_gather_rows(t.view(1, -1), buf[:need].view(self.world, -1))
        k.tree_sum(buf, self.world, t.numel(), t)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)

    def all_gather_rows(self, local, out):
        """out[(rank*n):(rank+1)*n] = local over all ranks (row blocks of equal size)."""
        if self.world == 1:
            out.copy_(local)
            return
        if self.dist.get_backend(self.group) == "nccl":       # RCCL: one fused all-gather
            self.dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        else:                                                   # gloo (tests): list form, row blocks are views of `out`
            self.dist.all_gather(list(out.chunk(self.world, dim=0)), local.contiguous(), group=self.group)


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


# ======================================================================================================
# parameter store: three Adam groups as flat device buffers, reference (Theano) layouts and names
# ======================================================================================================
class ParamGroup:
    def __init__(self, torch, names, shapes, device):
        self.names = list(names)
        self.offsets = {}
        o = 0
        for n in self.names:
            cnt = int(np.prod(shapes[n]))
            self.offsets[n] = (o, cnt, tuple(shapes[n]))
            o += (cnt + 3) // 4 * 4   # keep every tensor 16-byte aligned
        self.numel = o
        self.p = torch.zeros(o, dtype=torch.float32, device=device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.t = 0

    def view(self, buf, name):
        o, cnt, shape = self.offsets[name]
        return buf[o:o + cnt]


class BN:
    """Lasagne batch_norm in training mode (App. B.3): state of one normalisation in one pass."""

    def __init__(self, torch, C, device):
        self.C = C
        z = lambda: torch.zeros(C, dtype=torch.float32, device=device)
        self.sums, self.bsums = torch.zeros(2 * C, dtype=torch.float32, device=device), torch.zeros(2 * C, dtype=torch.float32, device=device)
        self.mean, self.inv_std, self.scale, self.shift = z(), z(), z(), z()
        self.count = 1.0


class _WriteLog(set):
    """The set of parameters whose gradient has been written in this backward sweep; every insertion is also reported to
    the trainer, which uses the sequence to launch a gradient bucket's all-reduce right after its LAST writer."""

    def __init__(self, owner):
        set.__init__(self)
        self.owner = owner

    def add(self, name):
        set.add(self, name)
        self.owner._wrote((name,))

    def update(self, names):
        names = tuple(names)
        set.update(self, names)
        self.owner._wrote(names)


class Trainer:
    """update_gen / update_discrim of train_IAN.py on one GPU (or one rank of a data-parallel job)."""

    def __init__(self, config_path, params, batch, comm=None, exact=True, device="cuda", deconv_flip=True):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise IanTrainError("the training step needs an MI355X: libian has no CPU fallback")
        self.lib = load_train_library()
        self.dev = torch.device(device)
        mod = config_loader.load_config(config_path)
        self.cfg = dict(mod.cfg)
        c = self.cfg
        self.n = int(batch)                      # per-rank batch
        self.comm = comm or Comm()
        self.exact = bool(exact) and self.comm.world > 1
        self.N = self.n * self.comm.world        # global batch (losses are means over it)
        self.k = K(self.lib, 0)
        self.lr = float(c["learning_rate"][0] if isinstance(c["learning_rate"], dict) else c["learning_rate"])
        self.zdim = int(c["num_latents"])
        self._build_params(params)
        self._build_layers(deconv_flip)
        self._alloc()
        self._dirty = {"enc", "Z", "dec"}
        self._plans, self._works, self._fired, self._buckets = {}, [], set(), None
        import collections
        self._ev, self._evlog = 0, []
        self.overlap_log = collections.deque(maxlen=256)    # last few steps only (read by tests/test_gpu_dp.py); bounded
        self._plan_key = {}
        self.measure_exposed = False                     # bench.py: time the compute stream's wait on the gradient all-reduce
        self._exposed_events = collections.deque(maxlen=64)
        if (self.n & (self.n - 1)) or (self.comm.world & (self.comm.world - 1)):
            import logging
            logging.getLogger(__name__).warning(
                "per-rank batch %d x world %d is not a power of two: batch-norm statistics are still exact, but the partial-sum "
                "tree (ian_k_tree_sum) walks them serially and the N-rank step is no longer bit-identical to the 1-rank step",
                self.n, self.comm.world)
        self.head6 = True                        # RGB-Beta head: R / G_a / B_a forward in one pass (kernels_head.hip)
        self.overlap = True                      # data parallel: all-reduce gradient buckets while backward still runs
        self.side = torch.cuda.Stream() if self.comm.world > 1 else None
        self.touched = _WriteLog(self)
        self.update_running = True

    # ---------------------------------------------------------------------------------------------
    def _build_params(self, P):
        torch = self.torch
        shapes = {k: tuple(np.shape(v)) for k, v in P.items()}
        enc = ["enc_conv1.W", "enc_conv1.b"]
        for i in (2, 3, 4):
            enc += ["enc_conv%d.W" % i, "bnorm%d.beta" % i, "bnorm%d.gamma" % i]
        enc += ["minibatch_discrim.theta", "minibatch_discrim.log_weight_scale", "minibatch_discrim.b", "discrimi.W"]
        zp = ["enc_fc1.W", "bnorm_enc_fc1.beta", "bnorm_enc_fc1.gamma", "enc_mu.W", "mu_bnorm.beta", "mu_bnorm.gamma",
              "enc_logsigma.W", "ls_bnorm.beta", "ls_bnorm.gamma"]
        dec = ["l_dec_fc2.W", "l_dec_fc2.b"]
        for dc, ci, co, hw, blk, sc in DEC_STAGES:
            dec.append(dc + ".W")
            dec += [blk + "bnorm0.beta", blk + "bnorm0.gamma"] + mdcl_names(blk, sc)
            dec += [blk + "bnorm1.beta", blk + "bnorm1.gamma"] + mdcl_names(blk + "2", sc)
            dec += [blk + "bnorm2.beta", blk + "bnorm2.gamma"]
        dec += ["dec_conv4.W", "bnorm_dc4.beta", "bnorm_dc4.gamma"]
        for h in ("R", "G_a", "G_b", "B_a", "B_b"):
            dec += mdcl_names(h, HEAD_SCALES)
        self.groups = {"enc": ParamGroup(torch, enc, shapes, self.dev), "Z": ParamGroup(torch, zp, shapes, self.dev),
                       "dec": ParamGroup(torch, dec, shapes, self.dev)}
        # batch-norm running averages (not trainable; Lasagne BatchNormLayer alpha = 0.1, App. B.3): what the
        # deterministic graphs of API.py / sample_IAN.py normalise with after training
        bn_names = ["bnorm2", "bnorm3", "bnorm4", "bnorm_enc_fc1", "mu_bnorm", "ls_bnorm", "bnorm_dc4"]
        for dc, ci, co, hw, blk, sc in DEC_STAGES:
            bn_names += [blk + "bnorm%d" % j for j in range(3)]
        stat_names = [b + t for b in bn_names for t in (".mean", ".inv_std")]
        self.stats = ParamGroup(torch, stat_names, shapes, self.dev)
        for nme in stat_names:
            self.stats.view(self.stats.p, nme).copy_(torch.from_numpy(np.ascontiguousarray(P[nme], np.float32).ravel()))
        self.frozen = {k: np.asarray(v, np.float32) for k, v in P.items() if k.startswith("l_IAF_")}
        self.bn_alpha = 0.1
        self.where = {}
        for gname, g in self.groups.items():
            for nme in g.names:
                if nme not in P:
                    raise IanTrainError("missing parameter %s" % nme)
                self.where[nme] = gname
                g.view(g.p, nme).copy_(torch.from_numpy(np.ascontiguousarray(P[nme], np.float32).ravel()))
        # MADE x2: never trained (train_IAN.py:184-194) -> pre-masked constants (layers.py:671,703)
        masks = made.masks_once(self.zdim)
        w, b = [], []
        for m in ("l_IAF_mu", "l_IAF_ls"):
            for l, mk in zip(("_input", "_output_W", "_output_D"), masks):
                w.append(np.asarray(P[m + l + ".W"], np.float32) * mk)
                b.append(np.asarray(P[m + l + ".b"], np.float32))
        self.made_w = torch.from_numpy(np.stack(w)).to(self.dev).contiguous()
        self.made_b = torch.from_numpy(np.stack(b)).to(self.dev).contiguous()
        # l_dec_fc2 output is stored (H,W,C) while its bias is indexed (C,H,W) (App. B.6)
        Cc, Hh, Ww = 512, 4, 4
        idx = np.arange(Cc * Hh * Ww).reshape(Cc, Hh, Ww).transpose(1, 2, 0).ravel()      # hwc position -> chw index
        self.fc2_perm = torch.from_numpy(idx.astype(np.int32)).to(self.dev)
        inv = np.empty_like(idx)
        inv[idx] = np.arange(idx.size)
        self.fc2_inv = torch.from_numpy(inv.astype(np.int32)).to(self.dev)                 # chw index -> hwc position

    def P(self, name):
        g = self.groups[self.where[name]]
        return g.view(g.p, name)

    def G(self, name):
        g = self.groups[self.where[name]]
        return g.view(g.g, name)

    def _build_layers(self, flip):
        L = lambda *a, **kw: Layer(self.lib, *a, deconv_flip=flip, **kw)
        self.layers = {}
        cin = 3
        for i, w in enumerate(ENC_WIDTHS):
            self.layers["enc_conv%d" % (i + 1)] = (L(K_CONV, cin, w, 64 >> i, 64 >> i), ["enc_conv%d.W" % (i + 1)])
            cin = w
        self.layers["enc_fc1"] = (L(K_DENSE, 16384, 1000, flat=(1024, 4, 4)), ["enc_fc1.W"])
        self.layers["enc_mu"] = (L(K_DENSE, 1000, self.zdim), ["enc_mu.W"])
        self.layers["enc_logsigma"] = (L(K_DENSE, 1000, self.zdim), ["enc_logsigma.W"])
        self.layers["mb"] = (L(K_DENSE, 1024, 2500), None)   # weights = normalised theta (layers.py:494)
        self.layers["l_dec_fc2"] = (L(K_DENSE, self.zdim, 8192, unflat=(512, 4, 4)), ["l_dec_fc2.W"])
        for dc, ci, co, hw, blk, sc in DEC_STAGES:
            self.layers[dc] = (L(K_DECONV, ci, co, hw, hw), [dc + ".W"])
            self.layers[blk] = (L(K_MDC, co, co, 2 * hw, 2 * hw, scales=sc), mdcl_names(blk, sc))
            self.layers[blk + "2"] = (L(K_MDC, co, co, 2 * hw, 2 * hw, scales=sc), mdcl_names(blk + "2", sc))
        self.layers["dec_conv4"] = (L(K_DECONV, 128, 128, 32, 32), ["dec_conv4.W"])
        for h, ci in (("R", 128), ("G_a", 128), ("G_b", 2), ("B_a", 128), ("B_b", 4)):
            self.layers[h] = (L(K_MDC, ci, 2, 64, 64, scales=HEAD_SCALES), mdcl_names(h, HEAD_SCALES))

    def _alloc(self):
        torch = self.torch
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.dev)
        self.ws_stats = z(256 * 2 * 8192)
        self.ws_loss = z(1024 * 2)
        self.scalars = z(64)
        self.mb_W = z(1024 * 2500)
        self.mb_dW = z(1024 * 2500)
        self.mb_colscale = z(2500)
        self.tmp_vals = z(2048)

    # ---------------------------------------------------------------------------------------------
    # parameter refresh (after every optimiser update): reference layout -> kernel layouts
    # ---------------------------------------------------------------------------------------------
    def refresh_weights(self):
        """Repack (device-side gather) the layers whose parameters changed since the last call: update_gen moves
        decoder_params + Z_params, update_discrim encoder_params + Z_params (train_IAN.py:274-276), so each step needs
        to repack about half of the weights, not all of them."""
        k = self.k
        dirty = self._dirty
        if not dirty:
            return
        for name, (layer, pnames) in self.layers.items():
            if pnames is None or self.where[pnames[0]] not in dirty:
                continue
            layer.set_params([self.P(n) for n in pnames])
        if "enc" in dirty:
            k.mb_weight(self.P("minibatch_discrim.theta"), self.P("minibatch_discrim.log_weight_scale"), self.mb_W, self.mb_colscale,
                        1024, 2500)
            self.layers["mb"][0].set_params([self.mb_W])
        if "dec" in dirty:
            if not hasattr(self, "fc2_bias"):
                self.fc2_bias = self.torch.empty(8192, dtype=self.torch.float32, device=self.dev)
            k.gather(self.P("l_dec_fc2.b"), self.fc2_perm, self.fc2_bias, 8192)
        self._dirty = set()

    def autotune(self):
        """Pick tile shape / split-K / K-loop schedule per layer for this per-rank batch by timing the real launches on this
        GPU (untimed set-up work, like API.IAN's ian_autotune; summation order aside, results do not change)."""
        torch = self.torch
        need = self.n * 64 * 64 * 128                      # the largest activation of IAN.py (dec_conv4 output)
        a = torch.randn(need, dtype=torch.float32, device=self.dev)
        b = torch.randn(need, dtype=torch.float32, device=self.dev)
        self.refresh_weights()
        for name, (layer, pnames) in self.layers.items():
            layer.autotune(self.n, a, b)
        torch.cuda.synchronize()

    def mark_params_changed(self, *groups):
        """Call after writing parameter values behind the trainer's back (tests, checkpoint loading)."""
        self._dirty.update(groups or ("enc", "Z", "dec"))

    # ---------------------------------------------------------------------------------------------
    # building blocks
    # ---------------------------------------------------------------------------------------------
    def _chunks(self, rows):
        """Row chunks of ian_k_colstats: the chunk SIZE depends on the per-image extent only (one image, or 512 rows of
        one), never on the batch, so a rank's partial sums are the very partial sums the single-process step forms for
        the same images (rank-order-invariant batch statistics; include/ian_train.h)."""
        rpi = max(1, rows // self.n)
        return self.n * max(1, rpi // 512) if rows == self.n * rpi else min(256, rows)

    def _ws(self, rows, C):
        need = self._chunks(rows) * 2 * C
        if self.ws_stats.numel() < need:
            self.ws_stats = self.torch.zeros(need, dtype=self.torch.float32, device=self.dev)
        return self.ws_stats

    def _bn_forward(self, bn, y, a, rows, C, stride, gamma, beta, act, count_rows, running=None):
        """batch statistics over this pass (all ranks when exact) -> a = act(bn(y)).  ``running``: name of the
        BatchNormLayer whose running averages this pass updates (the pass that sees the real minibatch)."""
        k = self.k
        upd = running is not None and self.update_running
        rm = self.stats.view(self.stats.p, running + ".mean") if upd else None
        ri = self.stats.view(self.stats.p, running + ".inv_std") if upd else None
        if not self.exact:                               # no collective between the two stages: one fused second stage
            bn.count = float(count_rows)
            k.bn_stats_affine(y, rows, C, stride, self._ws(rows, C), self._chunks(rows), bn.sums, bn.count, BN_EPS, gamma, beta,
                              bn.mean, bn.inv_std, bn.scale, bn.shift, rm, ri, 1.0 - self.bn_alpha, self.bn_alpha)
            k.affine(y, a, bn.scale, bn.shift, rows, C, stride, act)
            return
        k.colstats(0, y, None, None, None, None, rows, C, stride, 0, self._ws(rows, C), self._chunks(rows), bn.sums)
        self.comm.all_reduce_sum_ordered(bn.sums, k)
        bn.count = float(count_rows * self.comm.world)
        k.bn_make_affine(bn.sums, bn.count, BN_EPS, gamma, beta, C, bn.mean, bn.inv_std, bn.scale, bn.shift)
        k.affine(y, a, bn.scale, bn.shift, rows, C, stride, act)
        if upd:
            for r, cur in ((rm, bn.mean), (ri, bn.inv_std)):                  # r = (1-alpha) r + alpha * batch
                k.axpy(1.0 - self.bn_alpha, r, r, C, 0)
                k.axpy(self.bn_alpha, cur, r, C, 1)

    def _bn_backward(self, bn, dA, a, y, dy, rows, C, stride, act, gname, bname, want_w):
        k = self.k
        if not self.exact:
            gb = self.G(bname) if want_w else None
            gg = self.G(gname) if want_w else None
            ab, ag = int(bname in self.touched), int(gname in self.touched)
            k.bn_bwd_stats(dA, a, y, bn.mean, bn.inv_std, rows, C, stride, act, self._ws(rows, C), self._chunks(rows), bn.bsums,
                           gb, ab, gg, ag)
            if want_w:
                self.touched.update((bname, gname))
            k.bn_bwd(dA, a, y, bn.mean, bn.inv_std, bn.scale, bn.bsums, bn.count, dy, rows, C, stride, act)
            return
        k.colstats(1, dA, a, y, bn.mean, bn.inv_std, rows, C, stride, act, self._ws(rows, C), self._chunks(rows), bn.bsums)
        self.comm.all_reduce_sum_ordered(bn.bsums, k)
        if want_w:
            # with exact statistics every rank already holds the GLOBAL dbeta/dgamma: pre-divide so that the
            # gradient all-reduce (a sum over ranks) restores them
            sc = 1.0 / self.comm.world
            self._acc(bname, bn.bsums[:C], sc)
            self._acc(gname, bn.bsums[C:], sc)
        k.bn_bwd(dA, a, y, bn.mean, bn.inv_std, bn.scale, bn.bsums, bn.count, dy, rows, C, stride, act)

    def _acc(self, pname, src, alpha=1.0):
        """grad[pname] (+)= alpha * src"""
        self.k.axpy(float(alpha), src, self.G(pname), src.numel(), int(pname in self.touched))
        self.touched.add(pname)

    def _wgrad(self, lname, x, dy, n):
        layer, pnames = self.layers[lname]
        acc = pnames[0] in self.touched
        layer.backward_weight(x, dy, n, [self.G(p) for p in pnames], accumulate=acc)
        self.touched.update(pnames)

    def _head_backward(self, x, dR, dG, dB, n, dx, want_w):
        """Backward of R / G_a / B_a (IAN.py:183-199) once all three seeds are final: dx = sum of the three data gradients,
        plus the weight gradients -- one contract-first pass, or the per-layer calls."""
        names = ("R", "G_a", "B_a")
        ls = [self.layers[nm] for nm in names]
        accs = [pn[0] in self.touched for _, pn in ls]
        if self.head6 and len(set(accs)) == 1 and ls[0][0].head6_backward(
                ls[1][0], ls[2][0], x, (dR, dG, dB), n, 32, dx=dx, dx_stride=128,
                dparams3=[[self.G(p) for p in pn] for _, pn in ls] if want_w else None, accumulate=accs[0]):
            if want_w:
                for _, pn in ls:
                    self.touched.update(pn)
            return
        for i, (nm, dy) in enumerate(zip(names, (dR, dG, dB))):
            if want_w:
                self._wgrad(nm, x, dy, n)
            self.layers[nm][0].backward_data(dy, n, dx, accumulate=i > 0)

    # ---------------------------------------------------------------------------------------------
    # encoder pass (IAN.py:71-110 + discriminator head :209-216), training mode
    # ---------------------------------------------------------------------------------------------
    def enc_alloc(self):
        torch, n = self.torch, self.n
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.dev)
        E = {"x": z(n, 64, 64, 32), "dx": z(n, 64, 64, 32)}
        for i, w in enumerate(ENC_WIDTHS):
            hw = 32 >> i
            E["a%d" % (i + 1)] = z(n, hw, hw, w)
            E["da%d" % (i + 1)] = z(n, hw, hw, w)
            if i > 0:
                E["y%d" % (i + 1)] = z(n, hw, hw, w)
                E["bn%d" % (i + 1)] = BN(torch, w, self.dev)
        E["feat"], E["dfeat"] = z(n, 1024), z(n, 1024)
        E["act"], E["dact"] = z(n, cs(2500)), z(n, cs(2500))
        E["act_all"] = z(self.N if self.exact else n, cs(2500))
        E["mb"], E["dmb"] = z(n, cs(1524)), z(n, cs(1524))
        E["dmb_all"] = z(self.N if self.exact else n, cs(1524))
        E["p"], E["loss"], E["dlogits"] = z(n, 3), z(n, 4), z(n, 4)
        return E

    def enc_forward(self, E, x_nchw, targets=(-1, -1), acc_target=0, running=False):
        """x_nchw: (n,3,64,64) device tensor.  targets: classes whose -log p is recorded in E['loss'][:, 0:2]."""
        k, n = self.k, self.n
        k.nchw_to_nhwc(x_nchw, E["x"], n, 4096, 3, 32)
        self.layers["enc_conv1"][0].forward(E["x"], n, E["a1"], bias=self.P("enc_conv1.b"), act=ACT["lrelu"])
        for i in (2, 3, 4):
            w, hw = ENC_WIDTHS[i - 1], 64 >> i
            self.layers["enc_conv%d" % i][0].forward(E["a%d" % (i - 1)], n, E["y%d" % i])
            self._bn_forward(E["bn%d" % i], E["y%d" % i], E["a%d" % i], n * hw * hw, w, w, self.P("bnorm%d.gamma" % i),
                             self.P("bnorm%d.beta" % i), ACT["lrelu"], n * hw * hw, running=("bnorm%d" % i) if running else None)
        k.globalpool(E["a4"], E["feat"], n, 16, 1024, 1024, 1024)
        self.layers["mb"][0].forward(E["feat"], n, E["act"], y_stride=cs(2500))
        row0 = 0
        if self.exact:
            self.comm.all_gather_rows(E["act"], E["act_all"])
            row0 = self.comm.rank * n
        else:
            E["act_all"] = E["act"]
        E["row0"] = row0
        k.mb_forward(E["act_all"], E["act_all"].shape[0], cs(2500), row0, n, 500, 5, self.P("minibatch_discrim.b"), E["feat"], 1024,
                     1024, E["mb"], cs(1524))
        k.disc_head(E["mb"], cs(1524), 1524, self.P("discrimi.W"), n, targets[0], targets[1], acc_target, E["p"], E["loss"])

    def enc_backward(self, E, ce, feature_seeded, want_w, want_dx):
        """ce = (target0, w0, target1, w1): dlogits = sum w_t (p - onehot(target_t)).  feature_seeded: da1..da4 already
        hold the feature-loss seeds (train_IAN.py:244).  want_w: accumulate encoder_params gradients."""
        k, n = self.k, self.n
        t0, w0, t1, w1 = ce
        k.disc_head_bwd(E["p"], self.P("discrimi.W"), 1524, n, t0, float(w0), t1, float(w1), E["dlogits"], E["dmb"], cs(1524))
        if want_w:
            k.disc_head_wgrad(E["mb"], cs(1524), 1524, n, E["dlogits"], self.G("discrimi.W"), int("discrimi.W" in self.touched))
            self.touched.add("discrimi.W")
            # db[k] = sum_b df[b,k] : column sums of dmb[:, 1024:1524]
            k.colstats(2, E["dmb"].view(-1)[1024:], None, None, None, None, n, 500, cs(1524), 0, self.ws_stats, min(256, n), self.tmp_vals)
            self._acc("minibatch_discrim.b", self.tmp_vals[:500])
        dmb_all = E["dmb"]
        if self.exact:
            self.comm.all_gather_rows(E["dmb"], E["dmb_all"])
            dmb_all = E["dmb_all"]
        k.mb_backward(E["act_all"], E["act_all"].shape[0], cs(2500), E["row0"], n, 500, 5, dmb_all.view(-1)[1024:], cs(1524), E["dact"],
                      cs(2500))
        k.grad_pass(E["dmb"], cs(1524), 0, E["dfeat"], None, 1024, n, 1024, 0, 0)           # direct path of the concat (:524)
        self.layers["mb"][0].backward_data(E["dact"], n, E["dfeat"], dx_stride=1024, accumulate=True)
        if want_w:
            self.layers["mb"][0].backward_weight(E["feat"], E["dact"], n, [self.mb_dW], accumulate=False)
            acc = int("minibatch_discrim.theta" in self.touched)
            k.mb_weight_bwd(self.P("minibatch_discrim.theta"), self.mb_colscale, self.mb_dW, self.G("minibatch_discrim.theta"),
                            self.G("minibatch_discrim.log_weight_scale"), 1024, 2500, acc)
            self.touched.update(("minibatch_discrim.theta", "minibatch_discrim.log_weight_scale"))
        k.globalpool_bwd(E["dfeat"], E["da4"], n, 16, 1024, 1024, 1024, int(feature_seeded))
        for i in (4, 3, 2):
            w, hw = ENC_WIDTHS[i - 1], 64 >> i
            da, a, y = E["da%d" % i], E["a%d" % i], E["y%d" % i]
            self._bn_backward(E["bn%d" % i], da, a, y, da, n * hw * hw, w, w, ACT["lrelu"], "bnorm%d.gamma" % i, "bnorm%d.beta" % i, want_w)
            if want_w:
                self._wgrad("enc_conv%d" % i, E["a%d" % (i - 1)], da, n)
            self.layers["enc_conv%d" % i][0].backward_data(da, n, E["da%d" % (i - 1)], accumulate=feature_seeded)
        # enc_conv1: bias + lrelu, no batch-norm (IAN.py:71-80)
        if want_w:
            k.colstats(2, E["da1"], E["a1"], None, None, None, n * 1024, 128, 128, ACT["lrelu"], self.ws_stats, 256, self.tmp_vals)
            self._acc("enc_conv1.b", self.tmp_vals[:128])
        k.bn_bwd(E["da1"], E["a1"], None, None, None, None, None, 1.0, E["da1"], n * 1024, 128, 128, ACT["lrelu"])
        if want_w:
            self._wgrad("enc_conv1", E["x"], E["da1"], n)
        if want_dx:
            self.layers["enc_conv1"][0].backward_data(E["da1"], n, E["dx"], accumulate=False)

    # ---------------------------------------------------------------------------------------------
    # latent path (IAN.py:114-128): enc_fc1 -> (mu, logsigma) -> z0 = mu + e^ls * eps -> IAF
    # ---------------------------------------------------------------------------------------------
    def z_alloc(self):
        torch, n = self.torch, self.n
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.dev)
        Z = {"y_fc1": z(n, 1024), "f": z(n, 1024), "df": z(n, 1024), "bn_fc1": BN(torch, 1000, self.dev)}
        for nm in ("mu", "ls"):
            Z["y_" + nm], Z[nm], Z["d" + nm] = z(n, 128), z(n, 128), z(n, 128)
            Z["bn_" + nm] = BN(torch, 100, self.dev)
        Z["z0"], Z["z"], Z["dz0"], Z["dz"], Z["kl"] = z(n, 128), z(n, 128), z(n, 128), z(n, 128), z(n, 100)
        return Z

    def z_forward(self, Zs, a4, eps):
        k, n = self.k, self.n
        self.layers["enc_fc1"][0].forward(a4, n, Zs["y_fc1"], y_stride=1024)
        self._bn_forward(Zs["bn_fc1"], Zs["y_fc1"], Zs["f"], n, 1000, 1024, self.P("bnorm_enc_fc1.gamma"), self.P("bnorm_enc_fc1.beta"),
                         ACT["relu"], n, running="bnorm_enc_fc1")
        for nm, ln, bn in (("mu", "enc_mu", "mu_bnorm"), ("ls", "enc_logsigma", "ls_bnorm")):
            self.layers[ln][0].forward(Zs["f"], n, Zs["y_" + nm], y_stride=128)
            self._bn_forward(Zs["bn_" + nm], Zs["y_" + nm], Zs[nm], n, 100, 128, self.P(bn + ".gamma"), self.P(bn + ".beta"), 0, n, running=bn)
        k.sample(Zs["mu"], Zs["ls"], eps, Zs["z0"], Zs["kl"], n, 100, 128, eps.shape[1])
        k.made_iaf(Zs["z0"], Zs["z"], self.made_w, self.made_b, n, 100, 128)

    def z_backward(self, Zs, a4):
        """Zs['dz'] = dL/dz (from the decoder) -> gradients of Z_params, including KL and the L2 penalty."""
        k, n = self.k, self.n
        k.made_iaf_bwd(Zs["z0"], Zs["dz"], Zs["dz0"], self.made_w, self.made_b, n, 100, 128)
        klw = 1.0 / (self.N * 100.0)        # d(-0.5*mean(...)) : factor folded in the kernel's formula
        k.sample_bwd(Zs["mu"], Zs["ls"], Zs["eps"], Zs["dz0"], Zs["dmu"], Zs["dls"], n, 100, 128, Zs["eps"].shape[1], klw)
        first = True
        for nm, ln, bn in (("mu", "enc_mu", "mu_bnorm"), ("ls", "enc_logsigma", "ls_bnorm")):
            d = Zs["d" + nm]
            self._bn_backward(Zs["bn_" + nm], d, None, Zs["y_" + nm], d, n, 100, 128, 0, bn + ".gamma", bn + ".beta", True)
            self._wgrad(ln, Zs["f"], d, n)
            self.layers[ln][0].backward_data(d, n, Zs["df"], dx_stride=1024, accumulate=not first)
            first = False
        self._bn_backward(Zs["bn_fc1"], Zs["df"], Zs["f"], Zs["y_fc1"], Zs["df"], n, 1000, 1024, ACT["relu"], "bnorm_enc_fc1.gamma",
                          "bnorm_enc_fc1.beta", True)
        self._wgrad("enc_fc1", a4, Zs["df"], n)

    # ---------------------------------------------------------------------------------------------
    # decoder pass (IAN.py:129-207), training mode
    # ---------------------------------------------------------------------------------------------
    def dec_alloc(self):
        torch, n = self.torch, self.n
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=self.dev)
        D = {"h0": z(n, 4, 4, 512), "dh0": z(n, 4, 4, 512)}
        for dc, ci, co, hw, blk, sc in DEC_STAGES:
            s = 2 * hw
            for nm in ("x", "a", "b", "c", "e", "h"):
                D[blk + "_" + nm] = z(n, s, s, co)
            for nm in ("dx", "da", "dc", "dh"):
                D[blk + "_" + nm] = z(n, s, s, co)
            for j in range(3):
                D[blk + "_bn%d" % j] = BN(torch, co, self.dev)
        D["y4"], D["h4"], D["dh4"], D["bn4"] = z(n, 64, 64, 128), z(n, 64, 64, 128), z(n, 64, 64, 128), BN(torch, 128, self.dev)
        for nm in ("R", "G", "B", "Ga", "Ba", "RG", "gR", "gG", "gB", "dRG", "dRt"):
            D[nm] = z(n, 64, 64, 32)
        D["xhat"], D["dxhat"], D["tmp_img"] = z(n, 3, 64, 64), z(n, 3, 64, 64), z(n, 3, 64, 64)
        D["dz"] = z(n, 128)
        return D

    def dec_forward(self, D, zbuf, running=False):
        k, n = self.k, self.n
        rn = (lambda nme: nme) if running else (lambda nme: None)
        lay = lambda nme: self.layers[nme][0]
        lay("l_dec_fc2").forward(zbuf, n, D["h0"], y_stride=8192, bias=self.fc2_bias, act=ACT["lrelu"])
        h = D["h0"]
        for dc, ci, co, hw, blk, sc in DEC_STAGES:
            s = 2 * hw
            rows = n * s * s
            g = lambda j, t: self.P("%sbnorm%d.%s" % (blk, j, t))
            lay(dc).forward(h, n, D[blk + "_x"])
            self._bn_forward(D[blk + "_bn0"], D[blk + "_x"], D[blk + "_a"], rows, co, co, g(0, "gamma"), g(0, "beta"), ACT["lrelu"], rows, rn(blk + "bnorm0"))
            lay(blk).forward(D[blk + "_a"], n, D[blk + "_b"])
            self._bn_forward(D[blk + "_bn1"], D[blk + "_b"], D[blk + "_c"], rows, co, co, g(1, "gamma"), g(1, "beta"), ACT["lrelu"], rows, rn(blk + "bnorm1"))
            lay(blk + "2").forward(D[blk + "_c"], n, D[blk + "_e"], res=D[blk + "_x"])       # ElemwiseSum (layers.py:415)
            self._bn_forward(D[blk + "_bn2"], D[blk + "_e"], D[blk + "_h"], rows, co, co, g(2, "gamma"), g(2, "beta"), ACT["lrelu"], rows, rn(blk + "bnorm2"))
            h = D[blk + "_h"]
        rows = n * 4096
        lay("dec_conv4").forward(h, n, D["y4"])
        self._bn_forward(D["bn4"], D["y4"], D["h4"], rows, 128, 128, self.P("bnorm_dc4.gamma"), self.P("bnorm_dc4.beta"), ACT["lrelu"], rows, rn("bnorm_dc4"))
        sg = ACT["sigmoid"]
        # R = sigmoid(MDCL(h4)), G_a, B_a (IAN.py:183-199): the three layers that read the 128-channel map, one pass over it
        if not (self.head6 and lay("R").head6_forward(lay("G_a"), lay("B_a"), D["h4"], n, D["R"], D["Ga"], D["Ba"], 32, (sg, 0, 0))):
            lay("R").forward(D["h4"], n, D["R"], act=sg)                                      # IAN.py:183-186
            lay("G_a").forward(D["h4"], n, D["Ga"])
            lay("B_a").forward(D["h4"], n, D["Ba"])
        lay("G_b").forward(D["R"], n, D["G"], res=D["Ga"], act=sg)                            # :187-196
        k.concat2(D["R"], 2, 32, D["G"], 2, 32, D["RG"], 32, rows)                            # :201
        lay("B_b").forward(D["RG"], n, D["B"], res=D["Ba"], act=sg)                           # :197-206
        k.beta(D["R"], D["G"], D["B"], D["xhat"], n, 4096, 32)                                # :207

    def dec_backward(self, D, zbuf, want_w, want_dz):
        """D['dxhat'] (NCHW) -> gradients of decoder_params (want_w) and D['dz'] (want_dz)."""
        k, n = self.k, self.n
        lay = lambda nme: self.layers[nme][0]
        rows = n * 4096
        sg = ACT["sigmoid"]
        k.beta_bwd(D["dxhat"], D["R"], D["G"], D["B"], D["gR"], D["gG"], D["gB"], n, 4096, 32, sg)
        # B = sigmoid(B_a(h4) + B_b([R,G]))
        if want_w:
            self._wgrad("B_b", D["RG"], D["gB"], n)
        lay("B_b").backward_data(D["gB"], n, D["dRG"])
        k.grad_pass(D["dRG"], 32, 0, D["gR"], D["R"], 32, rows, 2, sg, 1)
        k.grad_pass(D["dRG"], 32, 2, D["gG"], D["G"], 32, rows, 2, sg, 1)
        # G = sigmoid(G_a(h4) + G_b(R))
        if want_w:
            self._wgrad("G_b", D["R"], D["gG"], n)
        lay("G_b").backward_data(D["gG"], n, D["dRt"])
        k.grad_pass(D["dRt"], 32, 0, D["gR"], D["R"], 32, rows, 2, sg, 1)
        # R = sigmoid(R(h4))
        self._head_backward(D["h4"], D["gR"], D["gG"], D["gB"], n, D["dh4"], want_w)      # all three seeds are final here
        # dec_conv4 + bnorm_dc4 + lrelu
        self._bn_backward(D["bn4"], D["dh4"], D["h4"], D["y4"], D["dh4"], rows, 128, 128, ACT["lrelu"], "bnorm_dc4.gamma", "bnorm_dc4.beta", want_w)
        prev_h = D[DEC_STAGES[-1][4] + "_h"]
        if want_w:
            self._wgrad("dec_conv4", prev_h, D["dh4"], n)
        lay("dec_conv4").backward_data(D["dh4"], n, D[DEC_STAGES[-1][4] + "_dh"])
        for si in range(len(DEC_STAGES) - 1, -1, -1):
            dc, ci, co, hw, blk, sc = DEC_STAGES[si]
            s = 2 * hw
            r = n * s * s
            bnn = lambda j, t: "%sbnorm%d.%s" % (blk, j, t)
            dh, dx, da, dcg = D[blk + "_dh"], D[blk + "_dx"], D[blk + "_da"], D[blk + "_dc"]
            # h = lrelu(bn2(x + d)),  d = MDCL2(c)
            self._bn_backward(D[blk + "_bn2"], dh, D[blk + "_h"], D[blk + "_e"], dh, r, co, co, ACT["lrelu"], bnn(2, "gamma"), bnn(2, "beta"), want_w)
            if want_w:
                self._wgrad(blk + "2", D[blk + "_c"], dh, n)
            lay(blk + "2").backward_data(dh, n, dcg)
            self._bn_backward(D[blk + "_bn1"], dcg, D[blk + "_c"], D[blk + "_b"], dcg, r, co, co, ACT["lrelu"], bnn(1, "gamma"), bnn(1, "beta"), want_w)
            if want_w:
                self._wgrad(blk, D[blk + "_a"], dcg, n)
            lay(blk).backward_data(dcg, n, da)
            self._bn_backward(D[blk + "_bn0"], da, D[blk + "_a"], D[blk + "_x"], dx, r, co, co, ACT["lrelu"], bnn(0, "gamma"), bnn(0, "beta"), want_w)
            k.axpy(1.0, dh, dx, dx.numel(), 1)                                               # residual edge: dx += d(x+d)
            src = D["h0"] if si == 0 else D[DEC_STAGES[si - 1][4] + "_h"]
            if want_w:
                self._wgrad(dc, src, dx, n)
            lay(dc).backward_data(dx, n, D["dh0"] if si == 0 else D[DEC_STAGES[si - 1][4] + "_dh"])
        # l_dec_fc2: bias + lrelu
        k.bn_bwd(D["dh0"], D["h0"], None, None, None, None, None, 1.0, D["dh0"], n, 8192, 8192, ACT["lrelu"])
        if want_w:
            self._wgrad("l_dec_fc2", zbuf, D["dh0"], n)
            k.colstats(2, D["dh0"], None, None, None, None, n, 8192, 8192, 0, self.ws_stats, min(256, n), self.tmp_vals_big())
            k.gather(self.tmp_vals_big(), self.fc2_inv, self._fc2_db(), 8192)
            self._acc("l_dec_fc2.b", self._fc2_db())
        if want_dz:
            lay("l_dec_fc2").backward_data(D["dh0"], n, D["dz"], dx_stride=128)

    def tmp_vals_big(self):
        if not hasattr(self, "_tvb"):
            self._tvb = self.torch.zeros(2 * 8192, dtype=self.torch.float32, device=self.dev)
        return self._tvb

    def _fc2_db(self):
        if not hasattr(self, "_fdb"):
            self._fdb = self.torch.zeros(8192, dtype=self.torch.float32, device=self.dev)
        return self._fdb

    # ---------------------------------------------------------------------------------------------
    # the step
    # ---------------------------------------------------------------------------------------------
    def _ensure_passes(self):
        if hasattr(self, "EX"):
            return
        self.EX, self.EH, self.EG = self.enc_alloc(), self.enc_alloc(), self.enc_alloc()
        self.ZS = self.z_alloc()
        self.DZ, self.DG = self.dec_alloc(), self.dec_alloc()
        self.zgen = self.torch.zeros(self.n, 128, dtype=self.torch.float32, device=self.dev)
        self.zgen0 = self.torch.zeros(self.n, 128, dtype=self.torch.float32, device=self.dev)

    def forward(self, X, Z, eps, xhat_override=None, xgen_override=None):
        """The three passes of train_IAN.py:116-149.  X (n,3,64,64), Z (n,100), eps (n,100): device tensors.
        ``*_override`` (test hook): images fed to the encoder passes on X_hat / X_gen instead of the decoder outputs
        (the decoders still run).  The discriminator's |a_b - a_b'| kernels and the leaky-ReLU kinks make the
        gradients discontinuous in the activations, so a parity test feeds both implementations the SAME images."""
        k, n = self.k, self.n
        self._ensure_passes()
        self.refresh_weights()
        self.X = X
        self.enc_forward(self.EX, X, targets=(0, -1), acc_target=0, running=True)             # p_X vs p1
        self.ZS["eps"] = eps
        self.z_forward(self.ZS, self.EX["a4"], eps)
        self.dec_forward(self.DZ, self.ZS["z"], running=True)                                  # X_hat
        self.enc_forward(self.EH, self.DZ["xhat"] if xhat_override is None else xhat_override, targets=(0, 1), acc_target=1)  # p_X_hat
        k.grad_pass(Z, Z.shape[1], 0, self.zgen0, None, 128, n, 100, 0, 0)                     # (n,100) -> padded rows
        k.made_iaf(self.zgen0, self.zgen, self.made_w, self.made_b, n, 100, 128)               # {l_Z_IAF: Z} (:149)
        self.dec_forward(self.DG, self.zgen)                                                   # X_gen
        self.enc_forward(self.EG, self.DG["xhat"] if xgen_override is None else xgen_override, targets=(0, 2), acc_target=2)  # p_X_gen

    def metrics(self):
        """All scalar losses of train_IAN.py:169-250,279 as a dict (one device->host copy).  Means are over the GLOBAL batch."""
        k, n, N = self.k, self.n, self.N
        s = self.scalars
        s.zero_()
        k.sum_rows(self.EX["loss"], n, 4, 1.0 / N, s[0:4])      # [0] discrim_d_loss, [2] acc(p_X)
        k.sum_rows(self.EH["loss"], n, 4, 1.0 / N, s[4:8])      # [4] gen_recon_loss, [5] CE(p_X_hat,p2), [6] acc
        k.sum_rows(self.EG["loss"], n, 4, 1.0 / N, s[8:12])     # [8] gen_sample_loss, [9] CE(p_X_gen,p3), [10] acc
        k.sum_rows(self.ZS["kl"], n * 100, 1, -0.5 / (N * 100.0), s[12:13])
        k.pair_loss(self.DZ["xhat"], self.X, None, n * 3 * 4096, 1, 1, 0, 0.0, 0, self.ws_loss, 1024, 1.0 / (N * 3 * 4096.0), s[16:18])
        for i, w in enumerate(ENC_WIDTHS):
            cnt = (32 >> i) ** 2 * w
            k.pair_loss(self.EH["a%d" % (i + 1)], self.EX["a%d" % (i + 1)], None, n * cnt, 1, 1, 1, 0.0, 0, self.ws_loss, 1024,
                        1.0 / (N * cnt * 4.0), s[20 + 2 * i:22 + 2 * i])
        if self.comm.world > 1:
            self.comm.all_reduce_sum(s)
        v = s.cpu().numpy()
        m = {"discrim_d_loss": v[0], "gen_recon_loss": v[4], "gen_sample_loss": v[8], "discrim_g_loss": v[5] + v[9],
             "discrim_acc": (v[2] + v[6] + v[10]) / 3.0, "kl_div": v[12], "pixel_loss": v[16], "pixel_acc": 1.0 - v[17],
             "feature_loss": v[20] + v[22] + v[24] + v[26]}
        return {kk: float(vv) for kk, vv in m.items()}

    def backward(self, which):
        """Gradients of the update rules of train_IAN.py:253-273 for ``which`` in {'gen', 'discrim'} (Z_params always)."""
        k, n, N, c = self.k, self.n, self.N, self.cfg
        self._begin_backward(which)
        EX, EH, EG, DZ, DG, ZS = self.EX, self.EH, self.EG, self.DZ, self.DG, self.ZS
        gen = which == "gen"
        # ---- shared generator-side loss S = adv_gen + recon_weight*pixel + feature_weight*feature -----------------
        for i, w in enumerate(ENC_WIDTHS):                                                     # feature_loss seeds (:244)
            cnt = (32 >> i) ** 2 * w
            k.pair_loss(EH["a%d" % (i + 1)], EX["a%d" % (i + 1)], EH["da%d" % (i + 1)], n * cnt, 1, 1, 1,
                        c["feature_weight"] / (4.0 * N * cnt), 0, self.ws_loss, 1024, 0.0, self.scalars[40:42])
        self.enc_backward(EH, (0, c["agr_weight"] / N, -1, 0.0), True, False, True)            # gen_recon_loss (:247)
        k.pair_loss(DZ["xhat"], self.X, DZ["dxhat"], n * 3 * 4096, 1, 1, 0, c["recon_weight"] / (N * 3 * 4096.0), 0, self.ws_loss, 1024,
                    0.0, self.scalars[40:42])                                                  # pixel_loss (:169)
        k.nhwc_to_nchw(EH["dx"], 32, DZ["tmp_img"], n, 4096, 3)
        k.axpy(1.0, DZ["tmp_img"], DZ["dxhat"], DZ["dxhat"].numel(), 1)
        self.dec_backward(DZ, ZS["z"], gen, True)
        ZS["dz"] = DZ["dz"]
        self.z_backward(ZS, EX["a4"])
        if gen:
            self.enc_backward(EG, (0, c["ags_weight"] / N, -1, 0.0), False, False, True)       # gen_sample_loss (:248)
            k.nhwc_to_nchw(EG["dx"], 32, DG["dxhat"], n, 4096, 3)
            self.dec_backward(DG, self.zgen, True, False)
        else:
            # ---- discriminator loss, X_hat and X_gen constant (consider_constant, :253) --------------------------
            self.enc_backward(EX, (0, c["dd_weight"] / N, -1, 0.0), False, True, False)        # discrim_d_loss (:234)
            self.enc_backward(EH, (1, c["dg_weight"] / N, -1, 0.0), False, True, False)        # p_X_hat vs p2 (:228)
            self.enc_backward(EG, (2, c["dg_weight"] / N, -1, 0.0), False, True, False)        # p_X_gen vs p3

    def _regularizers(self, which):
        """train_IAN.py:211-221: L2 on the Z parameters, orthogonal penalty on the 4-D weights of the updated group."""
        k, c = self.k, self.cfg
        for nme in self.groups["Z"].names:
            if not nme.endswith(".beta"):
                k.axpy(2.0 * c["reg"], self.P(nme), self.G(nme), self.P(nme).numel(), 1)
        if "ortho" not in c:
            return
        grp = self.groups["dec" if which == "gen" else "enc"]
        for nme in grp.names:
            o, cnt, shape = grp.offsets[nme]
            if nme[-1] == "W" and len(shape) == 4:
                k.ortho(self.P(nme), self.G(nme), shape[0], shape[1], shape[2], float(c["ortho"]), self.tmp_vals)

    def _adam(self, gname):
        g = self.groups[gname]
        g.t += 1
        b1, b2 = float(self.cfg["beta1"]), 0.999
        a_t = self.lr * math.sqrt(1.0 - b2 ** g.t) / (1.0 - b1 ** g.t)
        self.k.adam(g.p, g.g, g.m, g.v, g.numel, a_t, b1, b2, 1e-8)
        self._dirty.add(gname)

    # ---------------------------------------------------------------------------------------------
    # gradient all-reduce overlapped with backward
    # ---------------------------------------------------------------------------------------------
    # The flat gradient buffer of a group is cut into buckets.  A bucket may be reduced as soon as the last kernel that
    # writes into it has been issued: encoder_params receive contributions from three encoder passes, decoder_params
    # from two decoder passes, so "last" is a property of the whole backward sweep of a step kind.  The first step of
    # each kind records the order of gradient writes (every write goes through ``self.touched``); from the second step
    # on, the moment a bucket's last write is issued an event is recorded on the compute stream, a side stream waits
    # for it and the bucket's all-reduce is issued there (RCCL runs it on its own stream behind the side stream), while
    # the compute stream goes on with the rest of backward.  Z_params' bucket(s) finish before the second decoder /
    # the encoder passes even start.  Before the regularisers and Adam the compute stream waits for all the works.
    def _begin_backward(self, which):
        self.touched = _WriteLog(self)
        self._which, self._ev, self._evlog, self._fired, self._works = which, 0, [], set(), []
        self._buckets = None
        # the recorded write order depends on these switches: a plan made under other settings is discarded
        key = (bool(self.head6), bool(self.exact), bool(self.update_running))
        if self._plan_key.get(which) != key:
            self._plans.pop(which, None)
            self._plan_key[which] = key
        plan = self._plans.get(which) if (self.comm.world > 1 and self.overlap) else None
        if plan is not None:
            self._buckets = plan
            for b in plan:
                if b["ready"] == 0:
                    self._fire(b)

    def _wrote(self, names):
        self._ev += 1
        self._evlog.append(names)
        if self._buckets is None:
            return
        for b in self._buckets:
            if b["id"] in self._fired:
                if any(nm in b["names"] for nm in names):
                    self._plans.pop(self._which, None)          # stale plan: the next step of this kind re-records it
                    raise IanTrainError("gradient of %s written after its bucket was handed to the all-reduce "
                                        "(write order changed since the plan was recorded; plan dropped)" % (names,))
            elif b["ready"] == self._ev:
                self._fire(b)

    def _make_plan(self, which):
        """Buckets of the groups this step kind updates, each with the index of its last gradient write."""
        last = {}
        for i, names in enumerate(self._evlog):
            for nm in names:
                last[nm] = i + 1
        step = max(1, self.comm.bucket_bytes // 4)
        plan = []
        for gname in (("dec" if which == "gen" else "enc"), "Z"):
            g = self.groups[gname]
            for o in range(0, g.numel, step):
                e = min(g.numel, o + step)
                names = [nm for nm, (po, cnt, _) in g.offsets.items() if po < e and po + cnt > o]
                plan.append({"id": (gname, o), "group": gname, "lo": o, "hi": e, "names": set(names),
                             "ready": max([last.get(nm, 0) for nm in names] + [0])})
        return plan

    def _fire(self, b):
        torch = self.torch
        view = self.groups[b["group"]].g[b["lo"]:b["hi"]]
        ev = torch.cuda.Event()
        # every libian kernel of the step is launched on the legacy default stream (K(lib, 0), Layer(..., stream=0)):
        # the event must be recorded THERE, whatever torch's current stream is
        ev.record(torch.cuda.default_stream())        # after the bucket's last writer on the compute stream
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            w = self.comm.all_reduce_sum(view, async_op=True)
        self._fired.add(b["id"])
        self._works.append(w)
        self.overlap_log.append({"which": self._which, "bucket": b["id"], "bytes": 4 * (b["hi"] - b["lo"]), "issued_at_write": self._ev})

    def _finish_allreduce(self, which):
        """After backward: reduce whatever has not been handed over yet, then make the compute stream wait."""
        if self.comm.world == 1:
            return
        if self._buckets is None:                      # first step of this kind (or overlap off): plan, then reduce all
            self._plans[which] = self._make_plan(which)
            self._buckets = self._plans[which]
        for b in self._buckets:
            if b["id"] not in self._fired:
                self._fire(b)
        if self._works:
            for rec in list(self.overlap_log)[-len(self._works):]:
                rec["writes_in_backward"] = self._ev
        torch = self.torch
        timed = self.dev.type == "cuda" and self.measure_exposed
        if timed:                                      # how long the COMPUTE stream stalls on communication: the part of the
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # all-reduce backward did not hide
            e0.record(torch.cuda.default_stream())
        with torch.cuda.stream(torch.cuda.default_stream()):
            for w in self._works:
                w.wait()                               # RCCL: the COMPUTE (default) stream waits; gloo: host wait
        if timed:
            e1.record(torch.cuda.default_stream())
            self._exposed_events.append((which, e0, e1))
        self._works = []
        self._buckets = None

    def step(self, which, X, Z, eps, return_metrics=True):
        upd = "dec" if which == "gen" else "enc"
        if self.dev.type == "cuda" and self.torch.cuda.current_stream() != self.torch.cuda.default_stream():
            raise IanTrainError("Trainer.step must be called on the default stream (its kernels are launched on stream 0)")
        self.forward(X, Z, eps)
        m = self.metrics() if return_metrics else None
        self.backward(which)
        self._finish_allreduce(which)
        self._regularizers(which)
        self._adam(upd)
        self._adam("Z")
        return m

    def allreduce_exposed_ms(self):
        """Mean stall of the compute stream on the gradient all-reduce per step kind (needs measure_exposed = True and a
        device synchronisation before the call)."""
        out = {}
        for which, e0, e1 in self._exposed_events:
            out.setdefault(which, []).append(e0.elapsed_time(e1))
        return {k: float(sum(v) / len(v)) for k, v in out.items()}

    def update_gen(self, X, Z, eps):
        """train_IAN.py:309-318 -> [gen_recon_loss, gen_sample_loss, pixel_loss, feature_loss, pixel_acc]"""
        m = self.step("gen", X, Z, eps)
        return [m[k] for k in ("gen_recon_loss", "gen_sample_loss", "pixel_loss", "feature_loss", "pixel_acc")]

    def update_discrim(self, X, Z, eps):
        """train_IAN.py:320-329 -> [discrim_g_loss, discrim_d_loss, discrim_acc, pixel_loss, pixel_acc]"""
        m = self.step("discrim", X, Z, eps)
        return [m[k] for k in ("discrim_g_loss", "discrim_d_loss", "discrim_acc", "pixel_loss", "pixel_acc")]

    # ---- checkpoints (GANcheckpoints.py format, Theano parameter names: train_IAN.py:563-569) -------------
    def state_dict(self):
        """Every parameter the inference graph needs, reference layout and names: trainable groups, batch-norm
        running averages (updated from the real-data pass, alpha 0.1) and the frozen MADE parameters."""
        out = self.params_numpy()
        flat = self.stats.p.cpu().numpy()
        for n, (o, cnt, shape) in self.stats.offsets.items():
            out[n] = flat[o:o + cnt].reshape(shape).copy()
        out.update(self.frozen)
        return out

    def save_weights(self, fname, metadata=None):
        from . import checkpoints
        meta = {"learning_rate": self.lr}
        meta.update(metadata or {})
        checkpoints.save_weights(fname, self.state_dict(), meta)

    # ---- introspection for tests ------------------------------------------------------------------------
    def grads_numpy(self, gname):
        g = self.groups[gname]
        flat = g.g.cpu().numpy()
        return {n: flat[o:o + cnt].reshape(shape).copy() for n, (o, cnt, shape) in g.offsets.items()}

    def params_numpy(self):
        out = {}
        for g in self.groups.values():
            flat = g.p.cpu().numpy()
            for n, (o, cnt, shape) in g.offsets.items():
                out[n] = flat[o:o + cnt].reshape(shape).copy()
        return out
