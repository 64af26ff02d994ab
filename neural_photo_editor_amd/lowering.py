"""Lowers a recorded Lasagne-style model dict (config_loader.py) to the fused op list of include/ian.h.

Three sub-graphs are lowered, matching the Theano functions the reference compiles:
  ENC  l_in -> l_Z_IAF (or l_Z when there is no IAF), deterministic  (API.py:50, sample_IAN.py:90)
  IAF  l_Z_IAF -> l_Z                                                 (sample_IAN.py:93)
  DEC  l_Z -> l_out                                                   (API.py:46)
Fusion rules (pure host logic, tested on CPU):
  * [Conv|Deconv|Dense|MDCL] -> BatchNorm -> Nonlinearity with single consumers becomes ONE op whose
    epilogue carries the folded batch-norm and the activation;
  * ElemwiseSum([x, f(...)]) feeding BatchNorm/Nonlinearity (layers.py:412-416, IAN.py:187-206) becomes
    f's op with x as the residual input added before the affine;
  * BatchNorm applied directly to a tensor with other consumers (MDBLOCK's bnorm0) is an AFFINE op;
  * Dense after a conv map / before a ReshapeLayer carries the (C,H,W) flatten geometry so that
    finalize can permute the weights instead of moving data (App. B.6);
  * TransposedConv2DLayer(crop=1) + two SliceLayer(1:) (IAN_simple.py:182-223, dnn=False) is the same
    map as DeconvLayer(crop=2) (verified in the oracle tests) and lowers to the same op.
"""
from __future__ import annotations

from . import config_loader as cl

OP_CONV5S2, OP_DECONV5S2, OP_MDC3, OP_DENSE, OP_AFFINE, OP_MADE_IAF, OP_BETA, OP_CONCAT = 1, 2, 3, 4, 5, 6, 7, 8
SEG_ENC, SEG_IAF, SEG_DEC = 0, 1, 2
ACTS = {"identity": 0, "relu": 1, "lrelu": 2, "elu": 3, "tanh": 4, "sigmoid": 5}


class LoweringError(ValueError):
    pass


class Op(dict):
    """One fused op; keys mirror ian_op_desc."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _act_code(nl):
    if nl.kind == "lrelu" and abs(nl.leakiness - 0.2) > 1e-12:
        raise LoweringError("only LeakyRectify(0.2) is implemented (got %r)" % nl.leakiness)
    if nl.kind not in ACTS:
        raise LoweringError("nonlinearity %s is not supported on this path" % nl.kind)
    return ACTS[nl.kind]


class Lowered:
    def __init__(self):
        self.ops = []
        self.slots = []  # (h, w, c)
        self.slot_names = {}
        self.x_slot = self.zpre_slot = self.z_slot = self.out_slot = -1
        self.num_latents = 0
        self.params = []  # ParamSpec needed by the lowered ops
        self.has_made = False

    def slot_by_name(self, name):
        return self.slot_names[name]


class _Lowerer:
    def __init__(self, model):
        self.model = model
        self.out = Lowered()
        self.memo = {}
        roots = [model["l_out"], model["l_Z"]]
        self.consumers = {}
        for l in cl.get_all_layers(roots):
            for i in l.input_layers:
                self.consumers[id(i)] = self.consumers.get(id(i), 0) + 1
        self.segment = SEG_ENC
        self.param_seen = set()

    # ---- helpers -------------------------------------------------------------------------
    def ncons(self, layer):
        return self.consumers.get(id(layer), 0)

    def new_slot(self, shape4, name):
        if len(shape4) == 4:
            _, c, h, w = shape4
        else:
            _, c = shape4
            h = w = 1
        self.out.slots.append((int(h), int(w), int(c)))
        idx = len(self.out.slots) - 1
        if name:
            base, k = name, 1
            while name in self.out.slot_names:
                k += 1
                name = "%s#%d" % (base, k)
            self.out.slot_names[name] = idx
        return idx

    def use_params(self, layer):
        for p in layer.params:
            if p.name not in self.param_seen:
                self.param_seen.add(p.name)
                self.out.params.append(p)

    def emit_op(self, **kw):
        op = Op(kind=0, segment=self.segment, src=-1, src2=-1, src3=-1, dst=-1, cin=0, cout=0, in_h=1, in_w=1, act=0,
                has_bias=0, flat=(0, 0, 0), unflat=(0, 0, 0), scales=[], name="", bn_name=None)
        op.update(kw)
        self.out.ops.append(op)
        return op

    # ---- peeling of BN / nonlinearity / slice wrappers -------------------------------------
    def peel(self, layer):
        """Returns (core, bn, act_layer, slices) where the wrappers around ``core`` can be fused into it."""
        slices = []
        cur = layer
        while isinstance(cur, cl.SliceLayer) and cur.axis in (2, 3) and self.ncons(cur.input_layer) == 1:
            slices.append(cur)
            cur = cur.input_layer
        act = bn = None
        # collapse Nonlinearity chains (batch_norm() re-applies an identity nonlinearity, App. B.3)
        while isinstance(cur, cl.NonlinearityLayer) and self.ncons(cur.input_layer) == 1:
            if cur.nonlinearity.kind != "identity":
                if act is not None:
                    break
                act = cur
            cur = cur.input_layer
        if isinstance(cur, cl.BatchNormLayer) and self.ncons(cur.input_layer) == 1 and \
                isinstance(cur.input_layer, (cl._ConvBase, cl.DenseLayer, cl.MDCLLayer, cl.ElemwiseSumLayer)):
            bn = cur
            cur = cur.input_layer
        return cur, bn, act, slices

    def fused_act(self, core, bn, act_layer):
        own = getattr(core, "nonlinearity", None)
        own_code = _act_code(own) if own is not None else 0
        outer = _act_code(act_layer.nonlinearity) if act_layer is not None else 0
        if own_code and (outer or bn is not None):
            raise LoweringError("layer %s has its own nonlinearity under a BatchNorm/Nonlinearity wrapper" % core.name)
        return own_code or outer

    # ---- main recursion ------------------------------------------------------------------------
    def emit(self, layer):
        key = id(layer)
        if key in self.memo:
            return self.memo[key]
        slot = self._emit(layer)
        self.memo[key] = slot
        return slot

    def _emit(self, layer):
        if isinstance(layer, cl.GaussianSampleLayer):  # deterministic: mu (layers.py:431-432)
            return self.emit(layer.input_layers[0])
        if isinstance(layer, cl.ReshapeLayer):
            return self.emit_dense_like(layer.input_layer, reshape=layer)
        if isinstance(layer, cl.IAFLayer):
            return self.emit_iaf(layer)
        if isinstance(layer, cl.ConcatLayer):
            return self.emit_concat(layer)
        core, bn, act, slices = self.peel(layer)
        if slices and not isinstance(core, cl.TransposedConv2DLayer):
            raise LoweringError("SliceLayer on %s is not supported" % core.kind)
        if isinstance(core, cl.Conv2DLayer):
            return self.emit_conv(layer, core, bn, act)
        if isinstance(core, (cl.DeconvLayer, cl.TransposedConv2DLayer)):
            return self.emit_deconv(layer, core, bn, act, slices)
        if isinstance(core, cl.DenseLayer):
            return self.emit_dense_like(layer, reshape=None)
        if isinstance(core, cl.MDCLLayer):
            return self.emit_mdc(layer, core, bn, act, residual=None)
        if isinstance(core, cl.ElemwiseSumLayer):
            return self.emit_sum(layer, core, bn, act)
        if isinstance(core, cl.BatchNormLayer):  # BN (+nonlinearity) on a shared tensor -> AFFINE
            return self.emit_affine(layer, core, act)
        if isinstance(core, cl.NonlinearityLayer):  # nonlinearity on a shared tensor
            if act is not None:
                raise LoweringError("two stacked nonlinearities on a shared tensor are not supported")
            return self.emit_affine(layer, None, core)
        raise LoweringError("cannot lower layer %r" % (core,))

    def bn_fields(self, bn):
        if bn is None:
            return None
        self.use_params(bn)
        return bn.name

    def emit_conv(self, layer, core, bn, act):
        if core.filter_size != (5, 5) or core.stride != (2, 2) or core.pad != (2, 2):
            raise LoweringError("only 5x5 stride-2 pad-2 convolutions are implemented (%s)" % core.name)
        if core.flip_filters:
            raise LoweringError("%s: flip_filters=True (true convolution) is not implemented" % core.name)
        src = self.emit(core.input_layer)
        _, cin, h, w = core.input_shape
        self.use_params(core)
        dst = self.new_slot(core.output_shape, core.name)
        self.emit_op(kind=OP_CONV5S2, src=src, dst=dst, cin=cin, cout=core.num_filters, in_h=h, in_w=w,
                     act=self.fused_act(core, bn, act), has_bias=int(core.b is not None), name=core.name,
                     bn_name=self.bn_fields(bn))
        return dst

    def emit_deconv(self, layer, core, bn, act, slices):
        if core.filter_size != (5, 5) or core.stride != (2, 2):
            raise LoweringError("only 5x5 stride-2 transposed convolutions are implemented (%s)" % core.name)
        if isinstance(core, cl.TransposedConv2DLayer):
            ok = core.crop == (1, 1) and sorted(s.axis for s in slices) == [2, 3] and all(
                isinstance(s.indices, slice) and (s.indices.start, s.indices.stop, s.indices.step) in ((1, None, None),)
                for s in slices)
            if not ok:
                raise LoweringError("TransposedConv2DLayer is supported only as crop=1 + [1:] slices (IAN_simple.py:182-223)")
        elif core.crop != (2, 2):
            raise LoweringError("DeconvLayer crop must be (2,2) (%s)" % core.name)
        if core.flip_filters:
            raise LoweringError("%s: flip_filters=True is not implemented" % core.name)
        src = self.emit(core.input_layer)
        _, cin, h, w = core.input_shape
        self.use_params(core)
        dst = self.new_slot((None, core.num_filters, 2 * h, 2 * w), core.name)
        self.emit_op(kind=OP_DECONV5S2, src=src, dst=dst, cin=cin, cout=core.num_filters, in_h=h, in_w=w,
                     act=self.fused_act(core, bn, act), has_bias=int(core.b is not None), name=core.name,
                     bn_name=self.bn_fields(bn))
        return dst

    def emit_dense_like(self, layer, reshape):
        core, bn, act, slices = self.peel(layer)
        if not isinstance(core, cl.DenseLayer) or slices:
            raise LoweringError("ReshapeLayer is supported only directly after a DenseLayer stack")
        src = self.emit(core.input_layer)
        ishape = core.input_shape
        flat = (ishape[1], ishape[2], ishape[3]) if len(ishape) == 4 else (0, 0, 0)
        nin = core.W.shape[0]
        unflat = (0, 0, 0)
        oshape = (None, core.num_units)
        if reshape is not None:
            tgt = reshape.output_shape
            if len(tgt) != 4 or tgt[1] * tgt[2] * tgt[3] != core.num_units:
                raise LoweringError("unsupported ReshapeLayer target %r" % (tgt,))
            unflat = (tgt[1], tgt[2], tgt[3])
            oshape = (None, tgt[1], tgt[2], tgt[3])
        self.use_params(core)
        dst = self.new_slot(oshape, core.name)
        self.emit_op(kind=OP_DENSE, src=src, dst=dst, cin=nin, cout=core.num_units,
                     act=self.fused_act(core, bn, act), has_bias=int(core.b is not None), name=core.name,
                     bn_name=self.bn_fields(bn), flat=flat, unflat=unflat)
        if reshape is not None:
            self.memo[id(layer)] = dst
        return dst

    def emit_mdc(self, layer, core, bn, act, residual):
        if len(core.scales) > 4:
            raise LoweringError("MDCL with more than 4 scales")
        src = self.emit(core.input_layer)
        _, cin, h, w = core.input_shape
        self.use_params(core)
        dst = self.new_slot(core.output_shape, layer.name if (residual is not None and getattr(layer, "name", None)) else core.name)
        self.emit_op(kind=OP_MDC3, src=src, src2=-1 if residual is None else residual, dst=dst, cin=cin,
                     cout=core.num_filters, in_h=h, in_w=w, act=_act_code(act.nonlinearity) if act else 0,
                     name=core.name, bn_name=self.bn_fields(bn), scales=list(core.scales))
        return dst

    def emit_sum(self, layer, core, bn, act):
        ins = core.input_layers
        if len(ins) != 2:
            raise LoweringError("ElemwiseSumLayer with %d inputs" % len(ins))
        # fuse the sum into the MDCL that is used only here; the other operand is the residual
        cand = [i for i in (1, 0) if isinstance(ins[i], cl.MDCLLayer) and self.ncons(ins[i]) == 1]
        if not cand:
            raise LoweringError("ElemwiseSumLayer needs an exclusively-owned MDCL operand")
        k = cand[0]
        res = self.emit(ins[1 - k])
        return self.emit_mdc(layer, ins[k], bn, act, residual=res)

    def emit_affine(self, layer, bn, act):
        inner = bn.input_layer if bn is not None else act.input_layer
        src = self.emit(inner)
        shp = inner.output_shape
        c = shp[1]
        dst = self.new_slot(shp, (bn.name if bn is not None else None))
        self.emit_op(kind=OP_AFFINE, src=src, dst=dst, cin=c, cout=c, in_h=shp[2] if len(shp) == 4 else 1,
                     in_w=shp[3] if len(shp) == 4 else 1, act=_act_code(act.nonlinearity) if act else 0,
                     name=(bn.name if bn is not None else "nonlin"), bn_name=self.bn_fields(bn))
        return dst

    def emit_iaf(self, layer):
        z, mu, ls = layer.input_layers
        if not (isinstance(mu, cl.MADE) and isinstance(ls, cl.MADE) and mu.input_layer is z and ls.input_layer is z):
            raise LoweringError("IAFLayer must combine z with two MADEs of z (IAN.py:127-128)")
        if not (mu.name.endswith("_mu") and ls.name.endswith("_ls") and mu.name[:-3] == ls.name[:-3]):
            raise LoweringError("MADE layers must be named <x>_mu / <x>_ls")
        src = self.emit(z)
        d = z.output_shape[1]
        if mu.hidden_sizes != [d] or ls.hidden_sizes != [d]:
            raise LoweringError("MADE hidden size must equal the latent size")
        self.use_params(mu)
        self.use_params(ls)
        self.out.has_made = True
        dst = self.new_slot(z.output_shape, layer.name)
        self.emit_op(kind=OP_MADE_IAF, src=src, dst=dst, cin=d, cout=d, name=mu.name[:-3])
        return dst

    def emit_concat(self, layer):
        ins = layer.input_layers
        if layer.axis != 1:
            raise LoweringError("ConcatLayer on axis %d" % layer.axis)
        if len(ins) == 3 and all(isinstance(i, cl.beta_layer) for i in ins):
            srcs = []
            for b in ins:  # beta_layer(SliceLayer(T,0:1,1), SliceLayer(T,1:2,1)), IAN.py:207
                a, bb = b.input_layers
                ok = isinstance(a, cl.SliceLayer) and isinstance(bb, cl.SliceLayer) and a.input_layer is bb.input_layer \
                    and a.axis == 1 and bb.axis == 1 and (a.indices.start, a.indices.stop) == (0, 1) \
                    and (bb.indices.start, bb.indices.stop) == (1, 2) and a.input_layer.output_shape[1] == 2
                if not ok:
                    raise LoweringError("beta_layer inputs must be channel slices 0:1 / 1:2 of one 2-channel map")
                srcs.append(self.emit(a.input_layer))
            shp = ins[0].input_layers[0].input_layer.output_shape
            dst = self.new_slot((None, 3, shp[2], shp[3]), "l_out")
            self.emit_op(kind=OP_BETA, src=srcs[0], src2=srcs[1], src3=srcs[2], dst=dst, cin=2, cout=3, in_h=shp[2],
                         in_w=shp[3], name="beta")
            return dst
        if len(ins) == 2:
            a, b = self.emit(ins[0]), self.emit(ins[1])
            dst = self.new_slot(layer.output_shape, None)
            self.emit_op(kind=OP_CONCAT, src=a, src2=b, dst=dst, cin=ins[0].output_shape[1],
                         cout=layer.output_shape[1], in_h=layer.output_shape[2], in_w=layer.output_shape[3], name="concat")
            return dst
        raise LoweringError("unsupported ConcatLayer")

    # ---- driver -----------------------------------------------------------------------------------
    def run(self):
        m, out = self.model, self.out
        l_in, l_out, l_Z = m["l_in"], m["l_out"], m["l_Z"]
        l_Zpre = m.get("l_Z_IAF", l_Z)
        shp = l_in.output_shape
        if tuple(shp[1:]) != (3, 64, 64):
            raise LoweringError("this path is built for 3x64x64 images (got %r)" % (shp,))
        out.x_slot = self.new_slot(shp, "l_in")
        self.memo[id(l_in)] = out.x_slot
        self.segment = SEG_ENC
        out.zpre_slot = self.emit(l_Zpre)
        self.segment = SEG_IAF
        out.z_slot = self.emit(l_Z)
        # decoder: l_Z is an input (API.py:46 {l_Z: Z}); forget encoder-side memo of anything downstream
        self.segment = SEG_DEC
        out.out_slot = self.emit(l_out)
        out.num_latents = int(l_Z.output_shape[1])
        oh, ow, oc = out.slots[out.out_slot]
        if (oh, ow, oc) != (64, 64, 3):
            raise LoweringError("decoder output must be 3x64x64")
        return out


def lower_model(model):
    """model: dict returned by a config's get_model() under config_loader.stub_environment()."""
    for k in ("l_in", "l_out", "l_Z"):
        if k not in model:
            raise LoweringError("model dict lacks %r" % k)
    return _Lowerer(model).run()


def all_param_specs(model):
    """Every parameter of the model dict, in the spirit of API.py:25-29 (l_out + l_discrim + BN statistics)."""
    roots = [model["l_out"], model["l_Z"]]
    if "l_discrim" in model:
        roots.append(model["l_discrim"])
    return cl.get_all_params(roots)
