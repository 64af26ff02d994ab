"""TEST INFRASTRUCTURE -- puts the evaluating stand-ins (minitheano / minilasagne) into ``sys.modules`` under the
names the reference imports, adds /root/reference to ``sys.path`` and supplies the leftovers the
reference's hot-path files need under Python 3 / NumPy 2 (``xrange``, ``cPickle``, ``path.Path``, ``np.cast``).  The
reference's files are then imported UNMODIFIED:  ``with reference_modules() as ref: ref.API.IAN(...)``.
"""
from __future__ import annotations

import builtins
import contextlib
import importlib
import os
import pickle
import sys
import types
import warnings

from . import minilasagne as L
from . import minitheano as T

REFERENCE_DIR = os.environ.get("NPE_REFERENCE_DIR", "/root/reference")

_REF_MODULE_NAMES = ("layers", "mask_generator", "GANcheckpoints", "API", "IAN", "IAN_simple", "IANv1", "train_IAN",
                     "sample_IAN", "metrics_logging", "discgen_utils", "config")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def _public(module):
    return {k: v for k, v in vars(module).items() if not k.startswith("_")}


class _Path(str):
    """The sliver of path.py GANcheckpoints.py:24-30 uses."""

    def exists(self):
        return os.path.exists(self)

    def stripext(self):
        return _Path(os.path.splitext(self)[0])

    def rename(self, new):
        os.replace(self, new)
        return _Path(new)


def build_modules():
    mods = {}
    # ---- theano ---------------------------------------------------------------------------------
    nnet = _mod("theano.tensor.nnet", relu=L.nnet_relu, sigmoid=L.nnet_sigmoid, softmax=L.nnet_softmax,
                categorical_crossentropy=L.categorical_crossentropy)
    opt = _mod("theano.tensor.opt", register_canonicalize=lambda *a, **k: (a[0] if a else None))
    srs = _mod("theano.tensor.shared_randomstreams", RandomStreams=T.RandomStreams)
    tensor = _mod("theano.tensor", **_public(T))
    tensor.nnet, tensor.opt, tensor.shared_randomstreams = nnet, opt, srs
    rng_mrg = _mod("theano.sandbox.rng_mrg", MRG_RandomStreams=T.MRG_RandomStreams)
    basic_ops = _mod("theano.sandbox.cuda.basic_ops", as_cuda_ndarray_variable=L.as_cuda_ndarray_variable,
                     host_from_gpu=L.host_from_gpu, gpu_contiguous=L.gpu_contiguous, HostFromGpu=L.HostFromGpu,
                     gpu_alloc_empty=L.gpu_alloc_empty)
    cdnn = _mod("theano.sandbox.cuda.dnn", GpuDnnConvDesc=L.GpuDnnConvDesc, GpuDnnConv=L.GpuDnnConv,
                GpuDnnConvGradI=L.GpuDnnConvGradI, dnn_conv=L.dnn_conv, dnn_pool=L.dnn_pool)
    cuda = _mod("theano.sandbox.cuda", basic_ops=basic_ops, dnn=cdnn)
    sandbox = _mod("theano.sandbox", rng_mrg=rng_mrg, cuda=cuda)
    compile_ = _mod("theano.compile", SharedVariable=T.SharedVariable)
    config = types.SimpleNamespace(floatX=T.floatX)
    theano = _mod("theano", tensor=tensor, sandbox=sandbox, compile=compile_, config=config, shared=T.shared,
                  function=T.function, clone=T.clone, grad=T.grad, Variable=T.Variable)
    for m in (theano, tensor, nnet, opt, srs, sandbox, rng_mrg, cuda, basic_ops, cdnn, compile_):
        mods[m.__name__] = m
    # ---- lasagne --------------------------------------------------------------------------------
    init = _mod("lasagne.init", Initializer=L.Initializer, Normal=L.Normal, Constant=L.Constant,
                GlorotUniform=L.GlorotUniform, Orthogonal=L.Orthogonal)
    nonlin = _mod("lasagne.nonlinearities", identity=L.identity, linear=L.linear, rectify=L.rectify, elu=L.elu,
                  sigmoid=L.sigmoid, softmax=L.softmax, tanh=L.tanh, LeakyRectify=L.LeakyRectify,
                  leaky_rectify=L.leaky_rectify)
    utils = _mod("lasagne.utils", floatX=L.floatX, shared_empty=L.shared_empty, as_tuple=L.as_tuple, unique=L.unique,
                 collect_shared_vars=L.collect_shared_vars, create_param=L.create_param,
                 as_theano_expression=L.as_theano_expression)
    random_ = _mod("lasagne.random", get_rng=L.get_rng, set_rng=L.set_rng)
    updates = _mod("lasagne.updates", adam=L.adam, get_or_compute_grads=L.get_or_compute_grads)
    regul = _mod("lasagne.regularization", l1=L.l1, l2=L.l2, apply_penalty=L.apply_penalty,
                 regularize_layer_params=L.regularize_layer_params,
                 regularize_network_params=L.regularize_network_params)
    objectives = _mod("lasagne.objectives", squared_error=L.squared_error)
    layer_names = ("Layer MergeLayer InputLayer NonlinearityLayer SliceLayer ElemwiseMergeLayer ElemwiseSumLayer "
                   "ConcatLayer ReshapeLayer reshape GlobalPoolLayer PadLayer pad DenseLayer BatchNormLayer batch_norm "
                   "BaseConvLayer Conv2DLayer TransposedConv2DLayer Deconv2DLayer DilatedConv2DLayer Upscale2DLayer "
                   "get_all_layers get_output get_output_shape get_all_params get_all_param_values").split()
    layers = _mod("lasagne.layers", **{n: getattr(L, n) for n in layer_names})
    layers.dnn = _mod("lasagne.layers.dnn", Conv2DDNNLayer=L.Conv2DDNNLayer, Pool2DDNNLayer=L.Pool2DDNNLayer)
    layers.conv = _mod("lasagne.layers.conv", BaseConvLayer=L.BaseConvLayer, Conv2DLayer=L.Conv2DLayer,
                       TransposedConv2DLayer=L.TransposedConv2DLayer, DilatedConv2DLayer=L.DilatedConv2DLayer,
                       conv_output_length=L.conv_output_length, conv_input_length=L.conv_input_length)
    lasagne = _mod("lasagne", layers=layers, init=init, nonlinearities=nonlin, utils=utils, random=random_,
                   updates=updates, regularization=regul, objectives=objectives)
    for m in (lasagne, layers, layers.dnn, layers.conv, init, nonlin, utils, random_, updates, regul, objectives):
        mods[m.__name__] = m
    # ---- py2 / missing-dependency leftovers -----------------------------------------------------------
    mods["path"] = _mod("path", Path=_Path)
    mods["cPickle"] = pickle
    fuel_ds = _mod("fuel.datasets", CelebA=None)
    mods["fuel"] = _mod("fuel", datasets=fuel_ds)
    mods["fuel.datasets"] = fuel_ds
    return mods


class Reference(object):
    """Lazy attribute access to the reference's modules: ``ref.layers``, ``ref.API`` ..."""

    def __getattr__(self, name):
        if name not in _REF_MODULE_NAMES:
            raise AttributeError(name)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # py2-isms: `is 'W'` SyntaxWarning, imp deprecation
            return importlib.import_module(name)


@contextlib.contextmanager
def reference_modules(reference_dir=None):
    ref_dir = reference_dir or REFERENCE_DIR
    if not os.path.isdir(ref_dir):
        raise RuntimeError("reference checkout %s is not available (the fixtures under tests/golden/ref_*.npz are "
                           "generated in the build container only)" % ref_dir)
    mods = build_modules()
    saved = {k: sys.modules.get(k) for k in list(mods) + list(_REF_MODULE_NAMES)}
    had_xrange = hasattr(builtins, "xrange")
    sys.modules.update(mods)
    for n in _REF_MODULE_NAMES:
        sys.modules.pop(n, None)
    sys.path.insert(0, ref_dir)
    sys.dont_write_bytecode, old_dwb = True, sys.dont_write_bytecode  # /root/reference is read-only
    if not had_xrange:
        builtins.xrange = range
    import numpy as np

    class _Cast(dict):  # numpy<2 ``np.cast[dtype](x)`` (train_IAN.py:71), removed in NumPy 2.0
        def __missing__(self, dtype):
            return lambda x: np.asarray(x, dtype=dtype)
    had_cast = "cast" in vars(np)
    if not had_cast:
        np.cast = _Cast()
    try:
        yield Reference()
    finally:
        sys.dont_write_bytecode = old_dwb
        if not had_xrange:
            del builtins.xrange
        if not had_cast:
            del np.cast
        sys.path.remove(ref_dir)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
