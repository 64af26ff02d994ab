"""Executes a reference-style model config (IAN_simple.py / IAN.py: a Python file exposing
``cfg`` and ``get_model()``, API.py:18-21) WITHOUT Theano or Lasagne.

The config's ``import lasagne`` / ``import theano`` / ``from layers import ...`` statements are
satisfied by recording stand-ins installed in ``sys.modules`` for the duration of the load: every
layer constructor just records a graph node (type, hyper-parameters, parameter names and shapes
following SURVEY App. B.5).  ``get_model()`` therefore returns the same dict of "layers" the
reference returns (IAN_simple.py:235-241, IAN.py:219-228), which ``lowering.py`` turns into the
fused op list of ``include/ian.h``.  The reference's unmodified config files load through this.

Semantics that matter for the graph are restated from Lasagne [recalled]:
  * ``batch_norm(layer)`` removes the wrapped layer's bias and nonlinearity, inserts a
    BatchNormLayer named ``name`` and re-applies the nonlinearity (App. B.3) -- including when it is
    applied to an already-used layer, as MDBLOCK does (layers.py:412);
  * parameter names are ``<layer>.W``, ``<layer>.b``, ``<bn>.beta|gamma|mean|inv_std``; MDCL shared
    variables are ``<name>W``, ``<name>_coeff_*`` (layers.py:220,228,244,254).
"""
from __future__ import annotations

import contextlib
import importlib.util
import sys
import types

import numpy as np

# --------------------------------------------------------------------------------------
# nonlinearities / initialisers
# --------------------------------------------------------------------------------------


class Nonlinearity:
    def __init__(self, kind):
        self.kind = kind

    def __repr__(self):
        return "Nonlinearity(%s)" % self.kind

    def __call__(self, x):  # never evaluated symbolically
        raise TypeError("recorded nonlinearity is not callable")


class LeakyRectify(Nonlinearity):
    def __init__(self, leakiness=0.01):
        Nonlinearity.__init__(self, "lrelu")
        self.leakiness = leakiness


_NL = {k: Nonlinearity(k) for k in ("relu", "elu", "tanh", "sigmoid", "softmax", "identity")}


def _nl(x):
    """Normalise a nonlinearity argument (None == identity, BaseConvLayer/DenseLayer semantics)."""
    if x is None:
        return _NL["identity"]
    if isinstance(x, Nonlinearity):
        return x
    raise TypeError("unsupported nonlinearity %r" % (x,))


class Init:
    """lasagne.init stand-in; ``sample`` gives deterministic fallback values when a checkpoint lacks a
    parameter (the reference keeps its random initial value and warns, GANcheckpoints.py:53-54)."""

    def __init__(self, kind, *a, **kw):
        self.kind, self.a, self.kw = kind, a, kw

    def sample(self, shape, rs):
        k = self.kind
        if k == "Normal":
            std = self.a[0] if self.a else self.kw.get("std", 0.01)
            mean = self.a[1] if len(self.a) > 1 else self.kw.get("mean", 0.0)
            return rs.normal(mean, std, shape).astype(np.float32)
        if k == "Constant":
            val = self.a[0] if self.a else self.kw.get("val", 0.0)
            return np.full(shape, val, np.float32)
        if k == "Orthogonal":
            flat = (shape[0], int(np.prod(shape[1:])))
            a = rs.normal(0, 1, flat)
            u, _, v = np.linalg.svd(a, full_matrices=False)
            q = u if u.shape == flat else v
            gain = self.a[0] if self.a else self.kw.get("gain", 1.0)
            gain = np.sqrt(2.0) if gain == "relu" else float(gain)
            return (gain * q.reshape(shape)).astype(np.float32)
        if k == "GlorotUniform":
            fan = shape[0] + shape[1] if len(shape) == 2 else int(np.prod(shape[1:])) + shape[0]
            lim = np.sqrt(6.0 / fan)
            return rs.uniform(-lim, lim, shape).astype(np.float32)
        raise ValueError(k)


def _init_factory(kind):
    def make(*a, **kw):
        return Init(kind, *a, **kw)

    make.__name__ = kind
    return make


class ParamSpec:
    def __init__(self, name, shape, init, trainable=True):
        self.name, self.shape, self.init, self.trainable = name, tuple(int(s) for s in shape), init, trainable

    def __repr__(self):
        return "Param(%s,%s)" % (self.name, self.shape)


# --------------------------------------------------------------------------------------
# recorded layers
# --------------------------------------------------------------------------------------


def _pair(v):
    if isinstance(v, (tuple, list)):
        return (int(v[0]), int(v[1]))
    return (int(v), int(v))


class Layer:
    kind = "Layer"

    def __init__(self, incoming, name=None, **kwargs):
        if isinstance(incoming, (list, tuple)) and not (incoming and isinstance(incoming[0], (int, type(None)))):
            self.input_layers = list(incoming)
            self.input_layer = None
        else:
            self.input_layer = incoming if isinstance(incoming, Layer) else None
            self.input_layers = [self.input_layer] if self.input_layer is not None else []
        self.name = name
        self.params = []  # ParamSpec
        self.output_shape = None

    @property
    def input_shape(self):
        return self.input_layer.output_shape

    def add_param(self, suffix, shape, init, trainable=True, sep="."):
        p = ParamSpec("%s%s%s" % (self.name, sep, suffix), shape, init, trainable)
        self.params.append(p)
        return p

    def __repr__(self):
        return "<%s %s %s>" % (self.kind, self.name, self.output_shape)


class InputLayer(Layer):
    kind = "Input"

    def __init__(self, shape, input_var=None, name=None, **kw):
        Layer.__init__(self, None, name)
        self.output_shape = tuple(shape)
        self.shape = tuple(shape)


class _ConvBase(Layer):
    def _setup(self, incoming, num_filters, filter_size, stride, pad, W, b, nonlinearity, flip_filters, name):
        Layer.__init__(self, incoming, name)
        self.num_filters = int(num_filters)
        self.filter_size = _pair(filter_size)
        self.stride = _pair(stride)
        self.pad = pad if isinstance(pad, str) else _pair(pad)
        self.nonlinearity = _nl(nonlinearity)
        self.flip_filters = bool(flip_filters)
        self.W = self.add_param("W", self.get_W_shape(), W if isinstance(W, Init) else Init("Normal", 0.02))
        self.b = None if b is None else self.add_param("b", (self.num_filters,), b if isinstance(b, Init) else Init("Constant", 0.0))


class Conv2DLayer(_ConvBase):
    """lasagne.layers.Conv2DLayer (flip_filters defaults to True) [recalled]."""
    kind = "Conv2D"
    default_flip = True

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), pad=0, untie_biases=False,
                 W=None, b=Init("Constant", 0.0), nonlinearity=_NL["relu"], flip_filters=None, name=None, **kw):
        if flip_filters is None:
            flip_filters = self.default_flip
        self._setup(incoming, num_filters, filter_size, stride, pad, W, b, nonlinearity, flip_filters, name)
        n, c, hh, ww = self.input_shape
        ph, pw = self.pad
        self.output_shape = (n, self.num_filters, (hh + 2 * ph - self.filter_size[0]) // self.stride[0] + 1,
                             (ww + 2 * pw - self.filter_size[1]) // self.stride[1] + 1)

    def get_W_shape(self):
        return (self.num_filters, self.input_shape[1]) + self.filter_size


class Conv2DDNNLayer(Conv2DLayer):
    """lasagne.layers.dnn.Conv2DDNNLayer (flip_filters defaults to False) [recalled]."""
    default_flip = False


class TransposedConv2DLayer(_ConvBase):
    kind = "TransposedConv2D"

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), crop=0, untie_biases=False,
                 W=None, b=Init("Constant", 0.0), nonlinearity=_NL["relu"], flip_filters=False, name=None, **kw):
        self._setup(incoming, num_filters, filter_size, stride, crop, W, b, nonlinearity, flip_filters, name)
        self.crop = self.pad
        n, c, hh, ww = self.input_shape
        self.output_shape = (n, self.num_filters, (hh - 1) * self.stride[0] - 2 * self.crop[0] + self.filter_size[0],
                             (ww - 1) * self.stride[1] - 2 * self.crop[1] + self.filter_size[1])

    def get_W_shape(self):  # first two sizes swapped compared to a forward convolution
        return (self.input_shape[1], self.num_filters) + self.filter_size


class DeconvLayer(_ConvBase):
    """layers.py:436-483: output forced to exactly 2x the input (:460); W (Cin,Cout,kh,kw) (:449-452)."""
    kind = "Deconv"

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), crop=0, untie_biases=False,
                 W=None, b=Init("Constant", 0.0), nonlinearity=_NL["relu"], flip_filters=False, name=None, **kw):
        self._setup(incoming, num_filters, filter_size, stride, crop, W, b, nonlinearity, flip_filters, name)
        self.crop = self.pad
        n, c, hh, ww = self.input_shape
        self.output_shape = (n, self.num_filters, hh * 2, ww * 2)

    def get_W_shape(self):
        return (self.input_shape[1], self.num_filters) + self.filter_size


class DenseLayer(Layer):
    kind = "Dense"

    def __init__(self, incoming, num_units, W=None, b=Init("Constant", 0.0), nonlinearity=_NL["relu"], name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.num_units = int(num_units)
        self.nonlinearity = _nl(nonlinearity)
        nin = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param("W", (nin, self.num_units), W if isinstance(W, Init) else Init("GlorotUniform"))
        self.b = None if b is None else self.add_param("b", (self.num_units,), b if isinstance(b, Init) else Init("Constant", 0.0))
        self.output_shape = (self.input_shape[0], self.num_units)


class BatchNormLayer(Layer):
    kind = "BatchNorm"

    def __init__(self, incoming, axes="auto", epsilon=1e-4, alpha=0.1, name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.epsilon, self.alpha = epsilon, alpha
        feat = (self.input_shape[1],)
        self.beta = self.add_param("beta", feat, Init("Constant", 0.0))
        self.gamma = self.add_param("gamma", feat, Init("Constant", 1.0))
        self.mean = self.add_param("mean", feat, Init("Constant", 0.0), trainable=False)
        self.inv_std = self.add_param("inv_std", feat, Init("Constant", 1.0), trainable=False)
        self.output_shape = self.input_shape


class NonlinearityLayer(Layer):
    kind = "Nonlinearity"

    def __init__(self, incoming, nonlinearity=_NL["relu"], name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.nonlinearity = _nl(nonlinearity)
        self.output_shape = self.input_shape


def batch_norm(layer, **kwargs):
    """lasagne.layers.batch_norm [recalled], App. B.3."""
    nonlinearity = getattr(layer, "nonlinearity", None)
    if nonlinearity is not None:
        layer.nonlinearity = _NL["identity"]
    if getattr(layer, "b", None) is not None:
        layer.params = [p for p in layer.params if p is not layer.b]
        layer.b = None
    bn_name = kwargs.pop("name", None) or (getattr(layer, "name", None) and layer.name + "_bn")
    out = BatchNormLayer(layer, name=bn_name, **kwargs)
    if nonlinearity is not None:
        out = NonlinearityLayer(out, nonlinearity, name=bn_name and bn_name + "_nonlin")
    return out


class ElemwiseSumLayer(Layer):
    kind = "ElemwiseSum"

    def __init__(self, incomings, coeffs=1, cropping=None, name=None, **kw):
        Layer.__init__(self, list(incomings), name)
        self.output_shape = self.input_layers[0].output_shape


class ElemwiseMergeLayer(ElemwiseSumLayer):
    kind = "ElemwiseMerge"

    def __init__(self, incomings, merge_function=None, cropping=None, name=None, **kw):
        ElemwiseSumLayer.__init__(self, incomings, name=name)


class ConcatLayer(Layer):
    kind = "Concat"

    def __init__(self, incomings, axis=1, cropping=None, name=None, **kw):
        Layer.__init__(self, list(incomings), name)
        self.axis = axis
        shp = list(self.input_layers[0].output_shape)
        shp[axis] = sum(l.output_shape[axis] for l in self.input_layers)
        self.output_shape = tuple(shp)


class SliceLayer(Layer):
    kind = "Slice"

    def __init__(self, incoming, indices, axis=-1, name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.indices, self.axis = indices, axis
        shp = list(self.input_shape)
        ax = axis if axis >= 0 else len(shp) + axis
        if isinstance(indices, slice):
            if shp[ax] is not None:
                shp[ax] = len(range(*indices.indices(shp[ax])))
        else:
            del shp[ax]
        self.output_shape = tuple(shp)


class ReshapeLayer(Layer):
    kind = "Reshape"

    def __init__(self, incoming, shape, name=None, **kw):
        Layer.__init__(self, incoming, name)
        out = []
        for i, s in enumerate(shape):
            if isinstance(s, list):
                out.append(self.input_shape[s[0]])
            else:
                out.append(s)
        self.shape = tuple(shape)
        self.output_shape = tuple(out)


class GlobalPoolLayer(Layer):
    kind = "GlobalPool"

    def __init__(self, incoming, pool_function=None, name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.output_shape = self.input_shape[:2]


class PadLayer(Layer):
    kind = "Pad"

    def __init__(self, incoming, width, val=0, batch_ndim=2, name=None, **kw):
        Layer.__init__(self, incoming, name)
        self.width = width
        n, c, hh, ww = self.input_shape
        w = _pair(width)
        self.output_shape = (n, c, hh + 2 * w[0], ww + 2 * w[1])


# ---- the reference's own layers.py symbols that the configs import -------------------------------


class GaussianSampleLayer(Layer):
    """layers.py:419-433 (deterministic -> mu)."""
    kind = "GaussianSample"

    def __init__(self, mu, logsigma, rng=None, name=None, **kw):
        Layer.__init__(self, [mu, logsigma], name)
        self.output_shape = mu.output_shape


class IAFLayer(Layer):
    """layers.py:641-650."""
    kind = "IAF"

    def __init__(self, z, mu, logsigma, name=None, **kw):
        Layer.__init__(self, [z, mu, logsigma], name)
        self.output_shape = z.output_shape


class MADE(Layer):
    """layers.py:735-853, hidden_sizes=[h]: masked input layer, masked output layer + masked direct layer."""
    kind = "MADE"

    def __init__(self, z, hidden_sizes, name, nonlinearity=_NL["relu"], output_nonlinearity=None, **kw):
        Layer.__init__(self, z, name)
        self.hidden_sizes = list(hidden_sizes)
        d = z.output_shape[1]
        self.output_shape = z.output_shape
        if len(self.hidden_sizes) != 1:
            raise NotImplementedError("MADE with %d hidden layers" % len(self.hidden_sizes))
        hsz = self.hidden_sizes[0]
        orth = Init("Orthogonal", "relu")
        for sub, shp in (("_input", (d, hsz)), ("_output_W", (hsz, d)), ("_output_D", (d, d))):
            self.params.append(ParamSpec(name + sub + ".W", shp, orth))
            self.params.append(ParamSpec(name + sub + ".b", (shp[1],), Init("Constant", 0.0)))
        self.shuffled = None

    def reset(self, shuffling_type, last_shuffle=0):  # layers.py:845-853; masks come from made.py
        self.shuffled = shuffling_type

    def shuffle(self, shuffling_type):
        self.shuffled = shuffling_type


class MDCLLayer(Layer):
    """layers.py:207-258 recorded as one node (the reference builds it from 2-5 conv layers sharing W)."""
    kind = "MDCL"

    def __init__(self, incoming, num_filters, scales, name, dnn=True):
        Layer.__init__(self, incoming, name)
        self.num_filters, self.scales = int(num_filters), [int(s) for s in scales]
        ni = self.input_shape[1]
        sinit = Init("Constant", 1.0 / (1 + len(self.scales)))
        self.add_param("W", (self.num_filters, ni, 3, 3), Init("Normal", 0.02), sep="")
        self.add_param("_coeff_base", (self.num_filters,), sinit, sep="")
        for s in self.scales:
            self.add_param("_coeff_1x1" if s == 0 else "_coeff_%d" % s, (self.num_filters,), sinit, sep="")
        self.output_shape = (self.input_shape[0], self.num_filters) + tuple(self.input_shape[2:])


def MDCL(incoming, num_filters, scales, name, dnn=True):
    return MDCLLayer(incoming, num_filters, scales, name, dnn)


def MDBLOCK(incoming, num_filters, scales, name, nonlinearity):
    """layers.py:411-416."""
    a = NonlinearityLayer(batch_norm(incoming, name=name + "bnorm0"), nonlinearity)
    c = NonlinearityLayer(batch_norm(MDCL(a, num_filters, scales, name), name=name + "bnorm1"), nonlinearity)
    d = MDCL(c, num_filters, scales, name + "2")
    return NonlinearityLayer(batch_norm(ElemwiseSumLayer([incoming, d]), name=name + "bnorm2"), nonlinearity)


class beta_layer(Layer):
    """layers.py:397-408."""
    kind = "Beta"

    def __init__(self, alpha, beta, name=None, **kw):
        Layer.__init__(self, [alpha, beta], name)
        self.output_shape = alpha.output_shape


class MinibatchLayer(Layer):
    """layers.py:486-524 (discriminator only; recorded, not lowered for inference)."""
    kind = "Minibatch"

    def __init__(self, incoming, num_kernels, dim_per_kernel=5, name=None, **kw):
        Layer.__init__(self, incoming, name)
        nin = int(np.prod(self.input_shape[1:]))
        self.num_kernels, self.dim_per_kernel = num_kernels, dim_per_kernel
        self.add_param("theta", (nin, num_kernels, dim_per_kernel), Init("Normal", 0.05))
        self.add_param("log_weight_scale", (num_kernels, dim_per_kernel), Init("Constant", 0.0))
        self.add_param("b", (num_kernels,), Init("Constant", -1.0))
        self.output_shape = (self.input_shape[0], nin + num_kernels)


# --------------------------------------------------------------------------------------
# graph helpers (lasagne.layers.get_all_layers / get_all_params / get_output_shape)
# --------------------------------------------------------------------------------------


def get_all_layers(layer, treat_as_input=None):
    seen, order = set(), []
    stop = set(id(l) for l in (treat_as_input or []))

    def visit(l):
        if id(l) in seen:
            return
        seen.add(id(l))
        if id(l) not in stop:
            for i in l.input_layers:
                visit(i)
        order.append(l)

    for l in (layer if isinstance(layer, (list, tuple)) else [layer]):
        visit(l)
    return order


def get_all_params(layer, **tags):
    out = []
    for l in get_all_layers(layer):
        for p in l.params:
            if tags.get("trainable") is True and not p.trainable:
                continue
            if p not in out:
                out.append(p)
    return out


def get_output_shape(layer, input_shapes=None):
    if input_shapes is None or not isinstance(layer, Layer):
        return layer.output_shape
    shp = layer.output_shape
    return (input_shapes[0],) + tuple(shp[1:])


# --------------------------------------------------------------------------------------
# stand-in modules
# --------------------------------------------------------------------------------------


class _Any:
    """Permissive placeholder for every Theano/Lasagne name a config imports but never needs here."""

    def __init__(self, path="?"):
        self._path = path

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Any(self._path + "." + k)

    def __call__(self, *a, **kw):
        return _Any(self._path + "()")

    def __repr__(self):
        return "<stub %s>" % self._path


class _StubModule(types.ModuleType):
    def __init__(self, name, attrs=None):
        types.ModuleType.__init__(self, name)
        self.__path__ = []  # behave like a package so "import a.b.c" works
        for k, v in (attrs or {}).items():
            setattr(self, k, v)

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Any(self.__name__ + "." + k)


def _build_stub_modules():
    layer_names = dict(
        InputLayer=InputLayer, Conv2DLayer=Conv2DLayer, DenseLayer=DenseLayer, BatchNormLayer=BatchNormLayer,
        batch_norm=batch_norm, NonlinearityLayer=NonlinearityLayer, ElemwiseSumLayer=ElemwiseSumLayer,
        ElemwiseMergeLayer=ElemwiseMergeLayer, ConcatLayer=ConcatLayer, SliceLayer=SliceLayer,
        ReshapeLayer=ReshapeLayer, TransposedConv2DLayer=TransposedConv2DLayer, GlobalPoolLayer=GlobalPoolLayer,
        PadLayer=PadLayer, Layer=Layer, MergeLayer=Layer, get_all_layers=get_all_layers, get_all_params=get_all_params,
        get_output_shape=get_output_shape)
    mods = {}
    nl = _StubModule("lasagne.nonlinearities", dict(
        rectify=_NL["relu"], elu=_NL["elu"], tanh=_NL["tanh"], sigmoid=_NL["sigmoid"], softmax=_NL["softmax"],
        identity=_NL["identity"], linear=_NL["identity"], LeakyRectify=LeakyRectify))
    init = _StubModule("lasagne.init", {k: _init_factory(k) for k in ("Normal", "Constant", "Orthogonal", "GlorotUniform")})
    dnn = _StubModule("lasagne.layers.dnn", dict(Conv2DDNNLayer=Conv2DDNNLayer))
    conv = _StubModule("lasagne.layers.conv", dict(BaseConvLayer=_ConvBase))
    layers = _StubModule("lasagne.layers", dict(layer_names, dnn=dnn, conv=conv))
    lasagne = _StubModule("lasagne", dict(layers=layers, nonlinearities=nl, init=init))
    mods.update({"lasagne": lasagne, "lasagne.layers": layers, "lasagne.layers.dnn": dnn, "lasagne.layers.conv": conv,
                 "lasagne.nonlinearities": nl, "lasagne.init": init})
    for extra in ("lasagne.random", "lasagne.utils", "lasagne.updates", "lasagne.regularization", "lasagne.objectives"):
        m = _StubModule(extra)
        mods[extra] = m
        setattr(lasagne, extra.split(".")[1], m)
    theano_names = ["theano", "theano.tensor", "theano.tensor.shared_randomstreams", "theano.tensor.nnet",
                    "theano.sandbox", "theano.sandbox.rng_mrg", "theano.sandbox.cuda", "theano.sandbox.cuda.basic_ops",
                    "theano.sandbox.cuda.dnn", "theano.gpuarray", "theano.gpuarray.dnn"]
    for n in theano_names:
        mods[n] = _StubModule(n)
    for n in theano_names:
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(mods[parent], child, mods[n])
    mods["layers"] = _StubModule("layers", dict(
        GaussianSampleLayer=GaussianSampleLayer, MinibatchLayer=MinibatchLayer, DeconvLayer=DeconvLayer, MDCL=MDCL,
        MDBLOCK=MDBLOCK, beta_layer=beta_layer, MADE=MADE, IAFLayer=IAFLayer))
    mods["mask_generator"] = _StubModule("mask_generator")
    return mods


@contextlib.contextmanager
def stub_environment():
    mods = _build_stub_modules()
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def load_config(config_path):
    """imp.load_source('config', config_path) of API.py:18 under the recording stand-ins.
    Returns the executed module (``cfg``, ``get_model``)."""
    spec = importlib.util.spec_from_file_location("config", str(config_path))
    if spec is None or spec.loader is None:
        raise IOError("cannot load config %r" % (config_path,))
    module = importlib.util.module_from_spec(spec)
    with stub_environment():
        spec.loader.exec_module(module)
    return module


def build_model(config_module, dnn=True):
    """config_module.get_model(dnn=dnn) (API.py:21); IAN.py's get_model(interp=False) does not accept
    ``dnn``, so fall back to the no-argument call."""
    import inspect
    with stub_environment():
        try:
            params = inspect.signature(config_module.get_model).parameters
        except (TypeError, ValueError):
            params = {}
        if "dnn" in params:
            return config_module.get_model(dnn=dnn)
        return config_module.get_model()
