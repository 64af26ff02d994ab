"""TEST INFRASTRUCTURE -- evaluating stand-in for the slice of Lasagne (and of theano.tensor.nnet / the cuDNN
wrappers) the reference imports.  See ``minitheano.py`` for why this exists and who may import it.

Everything here restates PUBLISHED third-party behaviour from memory ("[recalled]"); it is the
only recalled part of a reference-executed fixture -- the compositions on top (MDCL, MDBLOCK, beta_layer,
MinibatchLayer, MADE/MaskedLayer/DIML, IAFLayer, GaussianSampleLayer, DeconvLayer's argument plumbing, the
model graphs, API.py's two gradients, train_IAN.py's loss/updates graph) are the reference's own lines.

Primitive conventions (NCHW):
  * Conv2DDNNLayer: flip_filters=False by default -> cuDNN 'cross' mode = correlation;
    Conv2DLayer: flip_filters=True by default -> true convolution; both ``pad`` symmetric zero padding.
  * TransposedConv2DLayer(flip_filters=False) = AbstractConv2d_gradInputs(filter_flip=True): the input gradient
    of a TRUE convolution; W is (in_channels, out_channels, kh, kw); output length (i-1)*s - 2*crop + k.
  * GpuDnnConvGradI with a conv_mode='conv' descriptor (layers.py:476-481): the same map, output shape given.
  * DilatedConv2DLayer: W is (in_channels, out_channels, kh, kw); out[b,f,y,x] = sum in[b,c,y+i*d,x+j*d] W[c,f,i,j]
    (AbstractConv2d_gradWeights(subsample=dilation, filter_flip=False) on batch/channel-swapped operands).
  * batch_norm(): strips bias and nonlinearity of the wrapped layer, BatchNormLayer(axes=all but 1,
    epsilon=1e-4, alpha=0.1), re-applies the nonlinearity.
  * nonlinearities: rectify = T.nnet.relu(x) = 0.5*(x+|x|); LeakyRectify(a) = relu(x, a) =
    0.5*(1+a)*x + 0.5*(1-a)*|x|; elu = switch(x>0, x, expm1(x)).
  * updates.adam: t<-t+1; a_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p <- p - a_t*m/(sqrt(v)+1e-8); every call
    owns t, m, v.
"""
from __future__ import annotations

import types
from collections import OrderedDict, deque

import numpy as np
import torch
import torch.nn.functional as F

from . import minitheano as T
from .minitheano import Variable, SharedVariable

floatX_name = "float32"


# ------------------------------------------------------------------------------------------------
# lasagne.utils / lasagne.random
# ------------------------------------------------------------------------------------------------
def floatX(arr):
    return np.asarray(arr, dtype=np.float32)


def shared_empty(dim=2, dtype=None):
    return T.shared(np.zeros((1,) * dim, dtype=dtype or floatX_name))


def as_theano_expression(x):
    if isinstance(x, (list, tuple)):
        return [as_theano_expression(e) for e in x]
    return T.as_tensor_variable(x)


def as_tuple(x, N, t=None):
    try:
        X = tuple(x)
    except TypeError:
        X = (x,) * N
    if t is not None and not all(isinstance(v, t) for v in X):
        raise TypeError("expected %s, got %r" % (t, x))
    if len(X) != N:
        raise ValueError("expected length %d, got %r" % (N, x))
    return X


def unique(l):
    out, seen = [], set()
    for el in l:
        if el not in seen:
            out.append(el)
            seen.add(el)
    return out


def collect_shared_vars(expressions):
    """Shared variables an expression (or list of) depends on, in graph order, without duplicates."""
    if isinstance(expressions, Variable):
        expressions = [expressions]
    return [v for v in T.ancestors(list(expressions)) if isinstance(v, SharedVariable)]


def create_param(spec, shape, name=None):
    if isinstance(spec, Variable):
        if shape is not None and spec.ndim != len(shape):
            raise ValueError("parameter variable has %d dimensions, should be %d" % (spec.ndim, len(shape)))
        if not spec.name:
            spec.name = name
        return spec
    shape = tuple(shape)
    if isinstance(spec, np.ndarray):
        if spec.shape != shape:
            raise ValueError("parameter array has shape %s, should be %s" % (spec.shape, shape))
        arr = spec
    elif callable(spec):
        arr = floatX(spec(shape))
        if arr.shape != shape:
            raise ValueError("cannot initialise parameter: wrong shape")
    else:
        raise TypeError("cannot create param from %r" % (spec,))
    return T.shared(arr, name=name)


_rng = np.random


def get_rng():
    return _rng


def set_rng(new_rng):
    global _rng
    _rng = new_rng


# ------------------------------------------------------------------------------------------------
# lasagne.init
# ------------------------------------------------------------------------------------------------
class Initializer(object):
    def __call__(self, shape):
        return self.sample(shape)


class Normal(Initializer):
    def __init__(self, std=0.01, mean=0.0):
        self.std, self.mean = std, mean

    def sample(self, shape):
        return floatX(get_rng().normal(self.mean, self.std, size=shape))


class Constant(Initializer):
    def __init__(self, val=0.0):
        self.val = val

    def sample(self, shape):
        return floatX(np.ones(shape) * self.val)


class GlorotUniform(Initializer):
    def __init__(self, gain=1.0, c01b=False):
        self.gain = np.sqrt(2) if gain == "relu" else gain

    def sample(self, shape):
        n1, n2 = shape[:2]
        rf = int(np.prod(shape[2:]))
        std = self.gain * np.sqrt(2.0 / ((n1 + n2) * rf))
        a = std * np.sqrt(3.0)
        return floatX(get_rng().uniform(-a, a, size=shape))


class Orthogonal(Initializer):
    def __init__(self, gain=1.0):
        self.gain = np.sqrt(2) if gain == "relu" else gain

    def sample(self, shape):
        flat = (shape[0], int(np.prod(shape[1:])))
        a = get_rng().normal(0.0, 1.0, flat)
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == flat else v
        return floatX(self.gain * q.reshape(shape))


# ------------------------------------------------------------------------------------------------
# theano.tensor.nnet / lasagne.nonlinearities
# ------------------------------------------------------------------------------------------------
def nnet_relu(x, alpha=0):
    if alpha == 0:
        return 0.5 * (x + abs(x))
    f1, f2 = 0.5 * (1 + alpha), 0.5 * (1 - alpha)
    return f1 * x + f2 * abs(x)


def nnet_sigmoid(x):
    x = T.as_tensor_variable(x)
    return Variable(lambda v: torch.sigmoid(v.to(T.F64)), [x], x.ndim, floatX_name)


def nnet_softmax(x):
    x = T.as_tensor_variable(x)
    return Variable(lambda v: torch.softmax(v.to(T.F64), dim=-1), [x], x.ndim, floatX_name)


def categorical_crossentropy(coding_dist, true_dist):
    """theano.tensor.nnet.categorical_crossentropy for a one-of-N matrix ``true_dist`` of the same rank."""
    if true_dist.ndim == coding_dist.ndim:
        return -T.sum(true_dist * T.log(coding_dist), axis=coding_dist.ndim - 1)
    raise NotImplementedError("integer-vector targets are not used by the reference")


def identity(x):
    return x


linear = identity
rectify = nnet_relu
sigmoid = nnet_sigmoid
softmax = nnet_softmax
tanh = T.tanh


def elu(x):
    return T.switch(x > 0, x, T.expm1(x))


class LeakyRectify(object):
    def __init__(self, leakiness=0.01):
        self.leakiness = leakiness

    def __call__(self, x):
        return nnet_relu(x, self.leakiness)


leaky_rectify = LeakyRectify()


# ------------------------------------------------------------------------------------------------
# lasagne.layers
# ------------------------------------------------------------------------------------------------
class Layer(object):
    def __init__(self, incoming, name=None):
        if isinstance(incoming, tuple):
            self.input_shape = incoming
            self.input_layer = None
        else:
            self.input_shape = incoming.output_shape
            self.input_layer = incoming
        self.name = name
        self.params = OrderedDict()
        self.get_output_kwargs = []
        if any(d is not None and d <= 0 for d in self.input_shape):
            raise ValueError("cannot create Layer with a non-positive input_shape dimension")

    @property
    def output_shape(self):
        shape = self.get_output_shape_for(self.input_shape)
        if any(isinstance(s, Variable) for s in shape):
            raise ValueError("%s returned a symbolic output shape" % type(self).__name__)
        return shape

    def get_params(self, unwrap_shared=True, **tags):
        result = list(self.params.keys())
        only = set(tag for tag, value in tags.items() if value)
        if only:
            result = [p for p in result if not (only - self.params[p])]
        exclude = set(tag for tag, value in tags.items() if not value)
        if exclude:
            result = [p for p in result if not (self.params[p] & exclude)]
        if unwrap_shared:
            return collect_shared_vars(result)
        return result

    def get_output_shape_for(self, input_shape):
        return input_shape

    def get_output_for(self, input, **kwargs):
        raise NotImplementedError

    def add_param(self, spec, shape, name=None, **tags):
        if name is not None and self.name is not None:
            name = "%s.%s" % (self.name, name)
        param = create_param(spec, shape, name)
        tags["trainable"] = tags.get("trainable", True)
        tags["regularizable"] = tags.get("regularizable", True)
        self.params[param] = set(tag for tag, value in tags.items() if value)
        return param


class MergeLayer(Layer):
    def __init__(self, incomings, name=None):
        self.input_shapes = [incoming if isinstance(incoming, tuple) else incoming.output_shape for incoming in incomings]
        self.input_layers = [None if isinstance(incoming, tuple) else incoming for incoming in incomings]
        self.name = name
        self.params = OrderedDict()
        self.get_output_kwargs = []

    @Layer.output_shape.getter
    def output_shape(self):
        return self.get_output_shape_for(self.input_shapes)


class InputLayer(Layer):
    def __init__(self, shape, input_var=None, name=None, **kwargs):
        self.shape = tuple(shape)
        ndim = len(shape)
        if input_var is None:
            input_var = T.TensorType(floatX_name, [s == 1 for s in shape])("input" if name is None else "%s.input" % name)
        self.input_var = input_var
        self.name = name
        self.params = OrderedDict()

    @Layer.output_shape.getter
    def output_shape(self):
        return self.shape


def get_all_layers(layer, treat_as_input=None):
    try:
        queue = deque(layer)
    except TypeError:
        queue = deque([layer])
    seen, done, result = set(), set(), []
    if treat_as_input is not None:
        seen.update(treat_as_input)
    while queue:
        layer = queue[0]
        if layer is None:
            queue.popleft()
        elif layer not in seen:
            seen.add(layer)
            if hasattr(layer, "input_layers"):
                queue.extendleft(reversed(layer.input_layers))
            elif hasattr(layer, "input_layer"):
                queue.appendleft(layer.input_layer)
        else:
            queue.popleft()
            if layer not in done:
                result.append(layer)
                done.add(layer)
    return result


def get_output(layer_or_layers, inputs=None, **kwargs):
    treat_as_input = list(inputs.keys()) if isinstance(inputs, dict) else []
    all_layers = get_all_layers(layer_or_layers, treat_as_input)
    all_outputs = dict((layer, layer.input_var) for layer in all_layers
                       if isinstance(layer, InputLayer) and layer not in treat_as_input)
    if isinstance(inputs, dict):
        all_outputs.update((layer, as_theano_expression(expr)) for layer, expr in inputs.items())
    elif inputs is not None:
        if len(all_outputs) > 1:
            raise ValueError("get_output() was called with a single input expression on a network with multiple input layers")
        for input_layer in all_outputs:
            all_outputs[input_layer] = as_theano_expression(inputs)
    for layer in all_layers:
        if layer not in all_outputs:
            try:
                if isinstance(layer, MergeLayer):
                    layer_inputs = [all_outputs[input_layer] for input_layer in layer.input_layers]
                else:
                    layer_inputs = all_outputs[layer.input_layer]
            except KeyError:
                raise ValueError("get_output() was called without giving an input expression for the free-floating "
                                 "layer %r" % layer)
            all_outputs[layer] = layer.get_output_for(layer_inputs, **kwargs)
    try:
        return [all_outputs[layer] for layer in layer_or_layers]
    except TypeError:
        return all_outputs[layer_or_layers]


def get_output_shape(layer_or_layers, input_shapes=None):
    if input_shapes is None or input_shapes == {}:
        try:
            return [layer.output_shape for layer in layer_or_layers]
        except TypeError:
            return layer_or_layers.output_shape
    treat_as_input = list(input_shapes.keys()) if isinstance(input_shapes, dict) else []
    all_layers = get_all_layers(layer_or_layers, treat_as_input)
    all_shapes = dict((layer, layer.shape) for layer in all_layers
                      if isinstance(layer, InputLayer) and layer not in treat_as_input)
    if isinstance(input_shapes, dict):
        all_shapes.update(input_shapes)
    else:
        for input_layer in all_shapes:
            all_shapes[input_layer] = input_shapes
    for layer in all_layers:
        if layer not in all_shapes:
            if isinstance(layer, MergeLayer):
                input_shapes_ = [all_shapes[l] for l in layer.input_layers]
            else:
                input_shapes_ = all_shapes[layer.input_layer]
            all_shapes[layer] = layer.get_output_shape_for(input_shapes_)
    try:
        return [all_shapes[layer] for layer in layer_or_layers]
    except TypeError:
        return all_shapes[layer_or_layers]


def get_all_params(layer, unwrap_shared=True, **tags):
    layers = get_all_layers(layer)
    params = []
    for l in layers:
        params.extend(l.get_params(unwrap_shared=unwrap_shared, **tags))
    return unique(params)


def get_all_param_values(layer, **tags):
    return [p.get_value() for p in get_all_params(layer, **tags)]


# -- simple layers ----------------------------------------------------------------------------------
class NonlinearityLayer(Layer):
    def __init__(self, incoming, nonlinearity=rectify, **kwargs):
        super(NonlinearityLayer, self).__init__(incoming, **kwargs)
        self.nonlinearity = identity if nonlinearity is None else nonlinearity

    def get_output_for(self, input, **kwargs):
        return self.nonlinearity(input)


class SliceLayer(Layer):
    def __init__(self, incoming, indices, axis=-1, **kwargs):
        super(SliceLayer, self).__init__(incoming, **kwargs)
        self.slice, self.axis = indices, axis

    def get_output_shape_for(self, input_shape):
        output_shape = list(input_shape)
        if isinstance(self.slice, int):
            del output_shape[self.axis]
        elif input_shape[self.axis] is not None:
            output_shape[self.axis] = len(range(*self.slice.indices(input_shape[self.axis])))
        return tuple(output_shape)

    def get_output_for(self, input, **kwargs):
        axis = self.axis
        if axis < 0:
            axis += input.ndim
        return input[(slice(None),) * axis + (self.slice,)]


class ElemwiseMergeLayer(MergeLayer):
    def __init__(self, incomings, merge_function, cropping=None, **kwargs):
        super(ElemwiseMergeLayer, self).__init__(incomings, **kwargs)
        self.merge_function = merge_function
        if cropping is not None:
            raise NotImplementedError("cropping")

    def get_output_shape_for(self, input_shapes):
        def match(dim1, dim2):
            if dim1 is not None and dim2 is not None and dim1 != dim2:
                raise ValueError("Mismatch: not all input shapes are the same: %r" % (input_shapes,))
            return dim1 if dim1 is not None else dim2
        out = input_shapes[0]
        for s in input_shapes[1:]:
            if len(s) != len(out):
                raise ValueError("Mismatch: not all input shapes have the same rank")
            out = tuple(match(a, b) for a, b in zip(out, s))
        return out

    def get_output_for(self, inputs, **kwargs):
        output = None
        for input in inputs:
            output = input if output is None else self.merge_function(output, input)
        return output


class ElemwiseSumLayer(ElemwiseMergeLayer):
    def __init__(self, incomings, coeffs=1, cropping=None, **kwargs):
        super(ElemwiseSumLayer, self).__init__(incomings, T.add, cropping=cropping, **kwargs)
        if isinstance(coeffs, list):
            if len(coeffs) != len(incomings):
                raise ValueError("Mismatch: got %d coeffs for %d incomings" % (len(coeffs), len(incomings)))
        else:
            coeffs = [coeffs] * len(incomings)
        self.coeffs = coeffs

    def get_output_for(self, inputs, **kwargs):
        inputs = [input * coeff if coeff != 1 else input for coeff, input in zip(self.coeffs, inputs)]
        return super(ElemwiseSumLayer, self).get_output_for(inputs, **kwargs)


class ConcatLayer(MergeLayer):
    def __init__(self, incomings, axis=1, cropping=None, **kwargs):
        super(ConcatLayer, self).__init__(incomings, **kwargs)
        self.axis = axis
        if cropping is not None:
            raise NotImplementedError("cropping")

    def get_output_shape_for(self, input_shapes):
        out = list(input_shapes[0])
        sizes = [s[self.axis] for s in input_shapes]
        out[self.axis] = None if any(s is None for s in sizes) else sum(sizes)
        for s in input_shapes[1:]:
            for ax, (a, b) in enumerate(zip(input_shapes[0], s)):
                if ax != self.axis % len(out) and a is not None and b is not None and a != b:
                    raise ValueError("Mismatch: input shapes must be the same except in the concatenation axis")
        return tuple(out)

    def get_output_for(self, inputs, **kwargs):
        return T.concatenate(inputs, axis=self.axis)


class ReshapeLayer(Layer):
    def __init__(self, incoming, shape, **kwargs):
        super(ReshapeLayer, self).__init__(incoming, **kwargs)
        self.shape = tuple(shape)

    def _resolve(self, input_shape):
        out = []
        for s in self.shape:
            if isinstance(s, list):
                out.append(("ref", s[0]))
            else:
                out.append(s)
        return out

    def get_output_shape_for(self, input_shape):
        out = [input_shape[s[0]] if isinstance(s, list) else s for s in self.shape]
        if -1 in out:
            known = [d for d in out if d != -1]
            if None not in known and None not in input_shape:
                out[out.index(-1)] = int(np.prod(input_shape)) // int(np.prod(known))
            else:
                out[out.index(-1)] = None
        return tuple(out)

    def get_output_for(self, input, **kwargs):
        shp = [input.shape[s[0]] if isinstance(s, list) else s for s in self.shape]
        return input.reshape(tuple(shp))


reshape = ReshapeLayer


class GlobalPoolLayer(Layer):
    def __init__(self, incoming, pool_function=T.mean, **kwargs):
        super(GlobalPoolLayer, self).__init__(incoming, **kwargs)
        self.pool_function = pool_function

    def get_output_shape_for(self, input_shape):
        return input_shape[:2]

    def get_output_for(self, input, **kwargs):
        return self.pool_function(input.flatten(3), axis=2)


class PadLayer(Layer):
    def __init__(self, incoming, width, val=0, batch_ndim=2, **kwargs):
        super(PadLayer, self).__init__(incoming, **kwargs)
        self.width, self.val, self.batch_ndim = width, val, batch_ndim

    def _widths(self, nd):
        w = self.width
        if isinstance(w, int):
            w = [w] * nd
        out = []
        for e in w:
            out.append((e, e) if isinstance(e, int) else tuple(e))
        return out

    def get_output_shape_for(self, input_shape):
        out = list(input_shape)
        for k, (l, r) in enumerate(self._widths(len(input_shape) - self.batch_ndim)):
            if out[k + self.batch_ndim] is not None:
                out[k + self.batch_ndim] += l + r
        return tuple(out)

    def get_output_for(self, input, **kwargs):
        widths = self._widths(input.ndim - self.batch_ndim)
        pads = []
        for l, r in reversed(widths):
            pads += [l, r]
        val = self.val
        return Variable(lambda v: F.pad(v, pads, value=float(val)), [input], input.ndim, input.dtype)


pad = PadLayer


class DenseLayer(Layer):
    def __init__(self, incoming, num_units, W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify,
                 num_leading_axes=1, **kwargs):
        super(DenseLayer, self).__init__(incoming, **kwargs)
        self.nonlinearity = identity if nonlinearity is None else nonlinearity
        self.num_units = num_units
        num_inputs = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param(W, (num_inputs, num_units), name="W")
        if b is None:
            self.b = None
        else:
            self.b = self.add_param(b, (num_units,), name="b", regularizable=False)

    def get_output_shape_for(self, input_shape):
        return (input_shape[0], self.num_units)

    def get_output_for(self, input, **kwargs):
        if input.ndim > 2:
            input = input.flatten(2)
        activation = T.dot(input, self.W)
        if self.b is not None:
            activation = activation + self.b.dimshuffle("x", 0)
        return self.nonlinearity(activation)


class BatchNormLayer(Layer):
    def __init__(self, incoming, axes="auto", epsilon=1e-4, alpha=0.1, beta=Constant(0), gamma=Constant(1),
                 mean=Constant(0), inv_std=Constant(1), **kwargs):
        super(BatchNormLayer, self).__init__(incoming, **kwargs)
        if axes == "auto":
            axes = (0,) + tuple(range(2, len(self.input_shape)))
        elif isinstance(axes, int):
            axes = (axes,)
        self.axes, self.epsilon, self.alpha = axes, epsilon, alpha
        shape = [size for axis, size in enumerate(self.input_shape) if axis not in self.axes]
        if any(size is None for size in shape):
            raise ValueError("BatchNormLayer needs specified input sizes for all axes not normalized over.")
        self.beta = None if beta is None else self.add_param(beta, shape, "beta", trainable=True, regularizable=False)
        self.gamma = None if gamma is None else self.add_param(gamma, shape, "gamma", trainable=True, regularizable=True)
        self.mean = self.add_param(mean, shape, "mean", trainable=False, regularizable=False)
        self.inv_std = self.add_param(inv_std, shape, "inv_std", trainable=False, regularizable=False)

    def get_output_for(self, input, deterministic=False, batch_norm_use_averages=None,
                       batch_norm_update_averages=None, **kwargs):
        input_mean = input.mean(self.axes)
        input_inv_std = T.inv(T.sqrt(input.var(self.axes) + self.epsilon))
        use_averages = deterministic if batch_norm_use_averages is None else batch_norm_use_averages
        if use_averages:
            mean, inv_std = self.mean, self.inv_std
        else:
            mean, inv_std = input_mean, input_inv_std
        update_averages = (not deterministic) if batch_norm_update_averages is None else batch_norm_update_averages
        if update_averages:
            running_mean = T.clone(self.mean, share_inputs=False)
            running_inv_std = T.clone(self.inv_std, share_inputs=False)
            running_mean.default_update = (1 - self.alpha) * running_mean + self.alpha * input_mean
            running_inv_std.default_update = (1 - self.alpha) * running_inv_std + self.alpha * input_inv_std
            mean = mean + 0 * running_mean
            inv_std = inv_std + 0 * running_inv_std
        param_axes = iter(range(input.ndim - len(self.axes)))
        pattern = ["x" if input_axis in self.axes else next(param_axes) for input_axis in range(input.ndim)]
        beta = 0 if self.beta is None else self.beta.dimshuffle(pattern)
        gamma = 1 if self.gamma is None else self.gamma.dimshuffle(pattern)
        mean = mean.dimshuffle(pattern)
        inv_std = inv_std.dimshuffle(pattern)
        return (input - mean) * (gamma * inv_std) + beta


def batch_norm(layer, **kwargs):
    nonlinearity = getattr(layer, "nonlinearity", None)
    if nonlinearity is not None:
        layer.nonlinearity = identity
    if hasattr(layer, "b") and layer.b is not None:
        del layer.params[layer.b]
        layer.b = None
    bn_name = kwargs.pop("name", None) or (getattr(layer, "name", None) and layer.name + "_bn")
    layer = BatchNormLayer(layer, name=bn_name, **kwargs)
    if nonlinearity is not None:
        nonlin_name = bn_name and bn_name + "_nonlin"
        layer = NonlinearityLayer(layer, nonlinearity, name=nonlin_name)
    return layer


# -- convolutions -----------------------------------------------------------------------------------
def conv_output_length(input_length, filter_size, stride, pad=0):
    if input_length is None:
        return None
    if pad == "valid":
        output_length = input_length - filter_size + 1
    elif pad == "full":
        output_length = input_length + filter_size - 1
    elif pad == "same":
        output_length = input_length
    elif isinstance(pad, int):
        output_length = input_length + 2 * pad - filter_size + 1
    else:
        raise ValueError("Invalid pad: %r" % (pad,))
    return (output_length + stride - 1) // stride


def conv_input_length(output_length, filter_size, stride, pad=0):
    if output_length is None:
        return None
    if pad == "valid":
        pad = 0
    elif pad == "full":
        pad = filter_size - 1
    elif pad == "same":
        pad = filter_size // 2
    if not isinstance(pad, int):
        raise ValueError("Invalid pad: %r" % (pad,))
    return (output_length - 1) * stride - 2 * pad + filter_size


def _conv2d(x, w, stride, pad, flip, dilation=(1, 1)):
    """conv of NCHW ``x`` with (out, in, kh, kw) ``w``; ``flip`` = true convolution."""
    def f(xv, wv):
        if flip:
            wv = torch.flip(wv, (2, 3))
        return F.conv2d(xv.to(T.F64), wv.to(T.F64), stride=stride, padding=pad, dilation=dilation)
    return Variable(f, [x, w], 4, floatX_name)


def _conv2d_grad_input(top, w, stride, pad, flip, out_hw):
    """Input gradient of ``_conv2d(bottom, w, stride, pad, flip)`` for a given bottom size (cuDNN GradI /
    AbstractConv2d_gradInputs).  ``w`` is (top_channels, bottom_channels, kh, kw)."""
    sym = [h for h in out_hw if isinstance(h, Variable)]

    def f(tv, wv, *s):
        it = iter(s)
        oh, ow = [int(next(it)) if isinstance(h, Variable) else int(h) for h in out_hw]
        if flip:
            wv = torch.flip(wv, (2, 3))
        kh, kw = wv.shape[2:]
        base_h = (tv.shape[2] - 1) * stride[0] - 2 * pad[0] + kh
        base_w = (tv.shape[3] - 1) * stride[1] - 2 * pad[1] + kw
        oph, opw = oh - base_h, ow - base_w
        if not (0 <= oph < stride[0] and 0 <= opw < stride[1]):
            raise ValueError("grad-input: requested size %s is inconsistent with the forward convolution" % ((oh, ow),))
        return F.conv_transpose2d(tv.to(T.F64), wv.to(T.F64), stride=stride, padding=pad, output_padding=(oph, opw))
    return Variable(f, [top, w] + sym, 4, floatX_name)


class BaseConvLayer(Layer):
    def __init__(self, incoming, num_filters, filter_size, stride=1, pad=0, untie_biases=False,
                 W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify, flip_filters=True, n=None, **kwargs):
        super(BaseConvLayer, self).__init__(incoming, **kwargs)
        self.nonlinearity = identity if nonlinearity is None else nonlinearity
        if n is None:
            n = len(self.input_shape) - 2
        elif n != len(self.input_shape) - 2:
            raise ValueError("Tried to create a %dD convolution layer with input shape %r." % (n, self.input_shape))
        self.n = n
        self.num_filters = num_filters
        self.filter_size = as_tuple(filter_size, n, int)
        self.flip_filters = flip_filters
        self.stride = as_tuple(stride, n, int)
        self.untie_biases = untie_biases
        if pad == "same":
            if any(s % 2 == 0 for s in self.filter_size):
                raise NotImplementedError("`same` padding requires odd filter size.")
        if pad == "valid":
            self.pad = as_tuple(0, n)
        elif pad in ("full", "same"):
            self.pad = pad
        else:
            self.pad = as_tuple(pad, n, int)
        self.W = self.add_param(W, self.get_W_shape(), name="W")
        if b is None:
            self.b = None
        else:
            if self.untie_biases:
                raise NotImplementedError("untie_biases")
            self.b = self.add_param(b, (num_filters,), name="b", regularizable=False)

    def get_W_shape(self):
        return (self.num_filters, self.input_shape[1]) + self.filter_size

    def get_output_shape_for(self, input_shape):
        pad = self.pad if isinstance(self.pad, tuple) else (self.pad,) * self.n
        return (input_shape[0], self.num_filters) + tuple(
            conv_output_length(i, f, s, p) for i, f, s, p in zip(input_shape[2:], self.filter_size, self.stride, pad))

    def get_output_for(self, input, **kwargs):
        conved = self.convolve(input, **kwargs)
        if self.b is None:
            activation = conved
        else:
            activation = conved + self.b.dimshuffle(("x", 0) + ("x",) * self.n)
        return self.nonlinearity(activation)

    def convolve(self, input, **kwargs):
        raise NotImplementedError("BaseConvLayer does not implement the convolve() method.")


def _int_pad(layer):
    if layer.pad == "full":
        return tuple(f - 1 for f in layer.filter_size)
    if layer.pad == "same":
        return tuple(f // 2 for f in layer.filter_size)
    return tuple(layer.pad)


class Conv2DLayer(BaseConvLayer):
    """theano.tensor.nnet.conv2d(filter_flip=flip_filters); flip_filters defaults to True."""

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), pad=0, untie_biases=False,
                 W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify, flip_filters=True, convolution=None, **kwargs):
        super(Conv2DLayer, self).__init__(incoming, num_filters, filter_size, stride, pad, untie_biases, W, b,
                                          nonlinearity, flip_filters, n=2, **kwargs)

    def convolve(self, input, **kwargs):
        return _conv2d(input, self.W, self.stride, _int_pad(self), self.flip_filters)


class Conv2DDNNLayer(BaseConvLayer):
    """lasagne.layers.dnn.Conv2DDNNLayer: conv_mode = 'conv' if flip_filters else 'cross'; flip_filters=False."""

    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), pad=0, untie_biases=False,
                 W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify, flip_filters=False, **kwargs):
        super(Conv2DDNNLayer, self).__init__(incoming, num_filters, filter_size, stride, pad, untie_biases, W, b,
                                             nonlinearity, flip_filters, n=2, **kwargs)

    def convolve(self, input, **kwargs):
        return _conv2d(input, self.W, self.stride, _int_pad(self), self.flip_filters)


class TransposedConv2DLayer(BaseConvLayer):
    def __init__(self, incoming, num_filters, filter_size, stride=(1, 1), crop=0, untie_biases=False,
                 W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify, flip_filters=False, output_size=None, **kwargs):
        super(TransposedConv2DLayer, self).__init__(incoming, num_filters, filter_size, stride, crop, untie_biases,
                                                    W, b, nonlinearity, flip_filters, n=2, **kwargs)
        self.crop = self.pad
        del self.pad
        self.output_size = output_size

    def get_W_shape(self):
        return (self.input_shape[1], self.num_filters) + self.filter_size

    def get_output_shape_for(self, input_shape):
        crop = getattr(self, "crop", getattr(self, "pad", None))
        crop = crop if isinstance(crop, tuple) else (crop,) * self.n
        return (input_shape[0], self.num_filters) + tuple(
            conv_input_length(i, f, s, p) for i, f, s, p in zip(input_shape[2:], self.filter_size, self.stride, crop))

    def convolve(self, input, **kwargs):
        crop = self.crop if isinstance(self.crop, tuple) else (self.crop,) * 2
        out_hw = self.get_output_shape_for(self.input_shape)[2:]
        # AbstractConv2d_gradInputs(filter_flip=not flip_filters)(W, input, output_size)
        return _conv2d_grad_input(input, self.W, self.stride, tuple(crop), not self.flip_filters, out_hw)


Deconv2DLayer = TransposedConv2DLayer


class DilatedConv2DLayer(BaseConvLayer):
    def __init__(self, incoming, num_filters, filter_size, dilation=(1, 1), pad=0, untie_biases=False,
                 W=GlorotUniform(), b=Constant(0.0), nonlinearity=rectify, flip_filters=False, **kwargs):
        self.dilation = as_tuple(dilation, 2, int)
        super(DilatedConv2DLayer, self).__init__(incoming, num_filters, filter_size, 1, pad, untie_biases, W, b,
                                                 nonlinearity, flip_filters, n=2, **kwargs)
        if self.pad != (0, 0):
            raise NotImplementedError("DilatedConv2DLayer requires pad=0 / (0,0) / 'valid', but got %r." % (pad,))
        if self.flip_filters:
            raise NotImplementedError("DilatedConv2DLayer does not support flip_filters=True")

    def get_W_shape(self):
        return (self.input_shape[1], self.num_filters) + self.filter_size

    def get_output_shape_for(self, input_shape):
        return (input_shape[0], self.num_filters) + tuple(
            conv_output_length(i, (f - 1) * d + 1, 1, 0)
            for i, f, d in zip(input_shape[2:], self.filter_size, self.dilation))

    def convolve(self, input, **kwargs):
        d = self.dilation

        def f(xv, wv):
            # out[b,f,y,x] = sum_c,i,j in[b,c,y+i*d,x+j*d] * W[c,f,i,j]
            return F.conv2d(xv.to(T.F64), wv.to(T.F64).permute(1, 0, 2, 3), dilation=d)
        return Variable(f, [input, self.W], 4, floatX_name)


class Upscale2DLayer(Layer):
    def __init__(self, *a, **k):
        raise NotImplementedError("Upscale2DLayer is imported by IAN.py but never instantiated")


class Pool2DDNNLayer(Layer):
    def __init__(self, *a, **k):
        raise NotImplementedError("Pool2DDNNLayer is imported by IAN.py but never instantiated")


# ------------------------------------------------------------------------------------------------
# theano.sandbox.cuda stand-ins used by layers.DeconvLayer.convolve (layers.py:467-483)
# ------------------------------------------------------------------------------------------------
def gpu_contiguous(x):
    return x


def as_cuda_ndarray_variable(x):
    return x


def host_from_gpu(x):
    return x


class HostFromGpu(object):
    def __call__(self, x):
        return x


def gpu_alloc_empty(*shape):
    class _Alloc(Variable):
        pass
    v = _Alloc(None, [], len(shape), floatX_name, name="gpu_alloc_empty")
    v.dims = shape  # ints or scalar Variables
    syms = [s for s in shape if isinstance(s, Variable)]
    v.inputs = syms

    def f(*vals):
        it = iter(vals)
        return torch.zeros([int(next(it)) if isinstance(s, Variable) else int(s) for s in shape], dtype=T.F64)
    v.fn = f
    return v


class _ConvDesc(object):
    def __init__(self, border_mode, subsample, conv_mode):
        self.border_mode, self.subsample, self.conv_mode = border_mode, subsample, conv_mode


class GpuDnnConvDesc(object):
    def __init__(self, border_mode, subsample=(1, 1), conv_mode="conv", precision=None):
        if conv_mode not in ("conv", "cross"):
            raise ValueError(conv_mode)
        self.border_mode, self.subsample, self.conv_mode = border_mode, tuple(subsample), conv_mode

    def __call__(self, img_shape, kern_shape):
        return _ConvDesc(self.border_mode, self.subsample, self.conv_mode)


class GpuDnnConvGradI(object):
    """cudnnConvolutionBackwardData: (kerns, topgrad, output-buffer, desc) -> gradient w.r.t. the forward input."""

    def __init__(self, inplace=False, workmem=None, algo=None):
        pass

    def __call__(self, kern, topgrad, output, desc, alpha=None, beta=None):
        bm = desc.border_mode
        pad = (bm, bm) if isinstance(bm, int) else tuple(bm)
        out_hw = output.dims[2:]
        return _conv2d_grad_input(topgrad, kern, tuple(desc.subsample), pad, desc.conv_mode == "conv", out_hw)


class GpuDnnConv(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("imported, never called by the reference")


def dnn_conv(*a, **k):
    raise NotImplementedError("imported, never called by the reference")


def dnn_pool(*a, **k):
    raise NotImplementedError("imported, never called by the reference")


# ------------------------------------------------------------------------------------------------
# lasagne.updates / regularization / objectives
# ------------------------------------------------------------------------------------------------
def get_or_compute_grads(loss_or_grads, params):
    if any(not isinstance(p, SharedVariable) for p in params):
        raise ValueError("params must contain shared variables only.")
    if isinstance(loss_or_grads, list):
        if not len(loss_or_grads) == len(params):
            raise ValueError("Got %d gradient expressions for %d parameters" % (len(loss_or_grads), len(params)))
        return loss_or_grads
    return T.grad(loss_or_grads, params)


def adam(loss_or_grads, params, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    all_grads = get_or_compute_grads(loss_or_grads, params)
    t_prev = T.shared(floatX(0.0))
    updates = OrderedDict()
    one = T.constant(1)
    t = t_prev + 1
    a_t = learning_rate * T.sqrt(one - beta2 ** t) / (one - beta1 ** t)
    for param, g_t in zip(params, all_grads):
        value = param.get_value(borrow=True)
        m_prev = T.shared(np.zeros(value.shape, dtype=value.dtype))
        v_prev = T.shared(np.zeros(value.shape, dtype=value.dtype))
        m_prev.adam_moment_of = (param, "m", beta1)  # fixture bookkeeping only (minitheano.Function keeps updates)
        v_prev.adam_moment_of = (param, "v", beta2)
        m_t = beta1 * m_prev + (one - beta1) * g_t
        v_t = beta2 * v_prev + (one - beta2) * g_t ** 2
        step = a_t * m_t / (T.sqrt(v_t) + epsilon)
        updates[m_prev] = m_t
        updates[v_prev] = v_t
        updates[param] = param - step
    t_prev.adam_step_counter = True
    updates[t_prev] = t
    return updates


def l1(x):
    return T.sum(abs(x))


def l2(x):
    return T.sum(x ** 2)


def apply_penalty(tensor_or_tensors, penalty, **kwargs):
    try:
        return sum(penalty(x, **kwargs) for x in tensor_or_tensors)
    except (TypeError, ValueError):
        return penalty(tensor_or_tensors, **kwargs)


def regularize_layer_params(layer, penalty, tags={"regularizable": True}, **kwargs):
    layers = [layer] if isinstance(layer, Layer) else layer
    all_params = []
    for l in layers:
        all_params += l.get_params(**tags)
    return apply_penalty(all_params, penalty, **kwargs)


def regularize_network_params(layer, penalty, tags={"regularizable": True}, **kwargs):
    return apply_penalty(get_all_params(layer, **tags), penalty, **kwargs)


def squared_error(a, b):
    return (a - b) ** 2


# ------------------------------------------------------------------------------------------------
# module assembly (see install.py)
# ------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m
