"""TEST INFRASTRUCTURE -- an *evaluating* stand-in for the slice of Theano the reference touches.

Purpose (VERDICT r2 item 1 / SURVEY 8c): the reference's own files (layers.py, mask_generator.py, IAN.py,
IAN_simple.py, API.py, train_IAN.py, GANcheckpoints.py under /root/reference) are imported UNMODIFIED on top
of this module and of ``minilasagne.py``; the golden vectors under tests/golden/ref_*.npz are what THEIR
lines compute.  Nothing here knows about IAN: it is a small lazy expression graph evaluated with torch-CPU in
float64 (the "truth" the float32 paths are compared against), with ``T.grad`` served by torch autograd.

What remains [recalled] (cannot be checked without the real libraries) and is therefore the
stated assumption of every reference-executed fixture:
  * the primitive conventions of the third-party ops (cuDNN conv / grad-input, AbstractConv2d_gradInputs,
    AbstractConv2d_gradWeights, see ``minilasagne.py``);
  * ``RandomStreams`` seeding: each random variable owns ``RandomState(int(RandomState(seed).randint(2**30)))``
    and ``seed()`` replays the sequence (theano/tensor/shared_randomstreams.py);
  * ``MRG_RandomStreams`` is NOT restated: ``multinomial`` only accepts one-hot rows (any correct sampler
    returns the row itself, which is all mask_generator.py:91 needs with l=0) and ``normal`` draws from a numpy
    RandomState and RECORDS every draw so the fixtures can hand the same epsilon to the code under test.

Only ``tests/golden/make_ref_golden.py`` imports this package.  It never travels to the GPU box (the fixtures do).
"""
from __future__ import annotations

import builtins
import warnings
from collections import OrderedDict

import numpy as np
import torch

warnings.filterwarnings('ignore', message='Converting a tensor with requires_grad=True to a scalar')
F64 = torch.float64
I64 = torch.int64
floatX = "float32"

_FLOATS = ("float32", "float64", "floatX")


def set_float_precision(bits):
    """64 (default): the float64 "truth".  32: evaluate the SAME graphs in float32 -- used only to measure how far a
    float32 evaluation of the reference's own graph sits from its float64 evaluation (the conditioning of a comparison)."""
    global F64
    F64 = {64: torch.float64, 32: torch.float32}[bits]


def _is_float(dt):
    return str(dt).startswith("float")


def _up(*dts):
    for d in dts:
        if _is_float(d):
            return floatX
    return "int64"


# ------------------------------------------------------------------------------------------------
# graph nodes
# ------------------------------------------------------------------------------------------------
class Variable(object):
    """A node: ``fn(*input_values) -> torch tensor``.  Identity-hashed (Theano does not overload ==)."""

    def __init__(self, fn=None, inputs=(), ndim=0, dtype=floatX, name=None):
        self.fn = fn
        self.inputs = list(inputs)
        self.ndim = int(ndim)
        self.dtype = dtype
        self.name = name
        self._subtensor = None  # (base, index) when produced by __getitem__ (inc_subtensor needs it)

    # -- python protocol ------------------------------------------------------------------------
    def __iter__(self):
        # theano/tensor/var.py: only vectors of known length iterate; everything the reference feeds to
        # lasagne.regularization.apply_penalty's first attempt (train_IAN.py:214-221) must raise TypeError.
        raise TypeError("TensorType does not support iteration.")

    def __bool__(self):
        raise TypeError("Variables do not support boolean operations.")

    def __repr__(self):
        return "<%s %s ndim=%d>" % (type(self).__name__, self.name, self.ndim)

    # -- arithmetic -----------------------------------------------------------------------------
    def __add__(self, o): return _ew(torch.add, self, o)
    def __radd__(self, o): return _ew(torch.add, o, self)
    def __sub__(self, o): return _ew(torch.sub, self, o)
    def __rsub__(self, o): return _ew(torch.sub, o, self)
    def __mul__(self, o): return _ew(torch.mul, self, o)
    def __rmul__(self, o): return _ew(torch.mul, o, self)
    def __truediv__(self, o): return _ew(_true_div, self, o, force_float=True)
    def __rtruediv__(self, o): return _ew(_true_div, o, self, force_float=True)
    __div__ = __truediv__
    __rdiv__ = __rtruediv__
    def __floordiv__(self, o): return _ew(lambda a, b: torch.div(a, b, rounding_mode="floor"), self, o)
    def __pow__(self, o): return _ew(torch.pow, self, o)
    def __rpow__(self, o): return _ew(torch.pow, o, self)
    def __neg__(self): return _ew1(torch.neg, self)
    def __abs__(self): return _ew1(torch.abs, self)
    def __lt__(self, o): return _cmp(torch.lt, self, o)
    def __le__(self, o): return _cmp(torch.le, self, o)
    def __gt__(self, o): return _cmp(torch.gt, self, o)
    def __ge__(self, o): return _cmp(torch.ge, self, o)

    # -- methods the reference calls ------------------------------------------------------------
    @property
    def shape(self):
        return Variable(lambda v: torch.tensor(list(v.shape), dtype=I64), [self], 1, "int64")

    @property
    def T(self):
        return transpose(self)

    @property
    def broadcastable(self):
        return (False,) * self.ndim

    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (list, tuple)):
            pattern = tuple(pattern[0])
        return dimshuffle(self, pattern)

    def flatten(self, outdim=1):
        def f(v):
            return v.reshape(tuple(v.shape[:outdim - 1]) + (-1,))
        return Variable(f, [self], outdim, self.dtype)

    def reshape(self, shape, ndim=None):
        return reshape(self, shape, ndim)

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (list, tuple)):
            axes = tuple(axes[0])
        return transpose(self, axes or None)

    def astype(self, dtype):
        return cast(self, dtype)

    def sum(self, axis=None, keepdims=False): return sum(self, axis, keepdims)
    def mean(self, axis=None, keepdims=False): return mean(self, axis, keepdims)
    def var(self, axis=None, keepdims=False): return var(self, axis, keepdims)
    def max(self, axis=None, keepdims=False): return max(self, axis, keepdims)
    def min(self, axis=None, keepdims=False): return min(self, axis, keepdims)

    def repeat(self, repeats, axis=None):
        def f(v, r):
            return torch.repeat_interleave(v, int(r), dim=axis)
        return Variable(f, [self, as_tensor_variable(repeats)], self.ndim, self.dtype)

    def eval(self, inputs_to_values=None):
        env = {}
        for k, v in (inputs_to_values or {}).items():
            env[k] = _to_torch(v, k.dtype)
        return _to_numpy(evaluate([self], env)[0], self.dtype)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        sym = []  # symbolic entries, in order of appearance

        def scan(e):
            if isinstance(e, Variable):
                sym.append(e)
            elif isinstance(e, slice):
                for p in (e.start, e.stop, e.step):
                    if isinstance(p, Variable):
                        sym.append(p)
        for e in idx:
            scan(e)
        nd = self.ndim
        for e in idx:
            if e is None:
                nd += 1
            elif isinstance(e, slice) or e is Ellipsis:
                pass
            elif isinstance(e, Variable) and e.ndim > 0:
                raise NotImplementedError("advanced indexing")
            else:
                nd -= 1

        def build(vals):
            it = iter(vals)

            def conc(p):
                return int(next(it)) if isinstance(p, Variable) else p
            out = []
            for e in idx:
                if isinstance(e, Variable):
                    out.append(int(next(it)))
                elif isinstance(e, slice):
                    out.append(slice(conc(e.start), conc(e.stop), conc(e.step)))
                elif isinstance(e, (int, np.integer)):
                    out.append(int(e))
                else:
                    out.append(e)
            return tuple(out)

        def f(v, *vals):
            ix = build(vals)
            # torch refuses negative steps: realise them with flip
            fixed, flips, dim = [], [], 0
            for e in ix:
                if isinstance(e, slice) and e.step is not None and e.step < 0:
                    if e.start is not None or e.stop is not None or e.step != -1:
                        raise NotImplementedError("general negative-step slice")
                    flips.append(dim)
                    fixed.append(slice(None))
                else:
                    fixed.append(e)
                if e is not None:
                    dim += 1
            if flips:
                v = torch.flip(v, flips)
            return v[tuple(fixed)]

        out = Variable(f, [self] + sym, nd, self.dtype)
        out._subtensor = (self, idx, sym, build)
        return out


class TensorPlaceholder(Variable):
    """A free input (T.matrix(), TensorType(...)('X'))."""

    def __init__(self, ndim, dtype, name=None):
        Variable.__init__(self, None, (), ndim, dtype, name)


class Constant(Variable):
    def __init__(self, value, dtype=None, name=None):
        arr = np.asarray(value)
        dt = dtype or (floatX if arr.dtype.kind == "f" else "int64")
        t = _to_torch(arr, dt)
        Variable.__init__(self, None, (), t.dim(), dt, name)
        self.value = t


class SharedVariable(Variable):
    """theano.compile.SharedVariable.  ``container`` is a 1-element list so ``clone()`` aliases storage exactly
    like SharedVariable.clone() does in Theano (the BatchNormLayer running-average trick relies on it)."""

    def __init__(self, value, name=None, container=None, dtype=None):
        if container is None:
            if isinstance(value, np.random.RandomState):
                container = [value]
                dtype, nd = "rng", 0
            else:
                arr = np.asarray(value)
                dtype = dtype or str(arr.dtype)
                container = [_to_torch(arr, dtype)]
                nd = arr.ndim
        else:
            nd = 0 if dtype == "rng" else container[0].dim()
        Variable.__init__(self, None, (), nd, dtype, name)
        self.container = container

    def get_value(self, borrow=False, return_internal_type=False):
        v = self.container[0]
        if self.dtype == "rng":
            return v
        return _to_numpy(v, self.dtype)

    def get_value_f64(self):
        """The un-rounded stored value (fixtures read this, not the float32 view)."""
        return self.container[0].detach().numpy().copy()

    def set_value(self, value, borrow=False):
        if self.dtype == "rng":
            self.container[0] = value
        else:
            self.container[0] = _to_torch(np.asarray(value), self.dtype)

    def clone(self):
        return SharedVariable(None, self.name, container=self.container, dtype=self.dtype)


def shared(value, name=None, borrow=False, broadcastable=None, strict=False, allow_downcast=None):
    return SharedVariable(value, name)


def _to_torch(x, dtype=None):
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        t = torch.from_numpy(np.array(a, copy=True, order='C')) if a.ndim else torch.tensor(a.item(), dtype=(F64 if a.dtype.kind == 'f' else (torch.bool if a.dtype.kind == 'b' else I64)))
    if t.dtype.is_floating_point or (dtype is not None and _is_float(dtype)):
        return t.to(F64)
    if t.dtype == torch.bool:
        return t.to(I64)
    return t.to(I64)


def _to_numpy(t, dtype=None):
    a = np.array(t.detach().numpy(), copy=True)
    if dtype is not None and dtype not in ("rng",) and str(a.dtype) != str(dtype):
        try:
            a = a.astype(dtype)
        except TypeError:
            pass
    return a


def as_tensor_variable(x, name=None, ndim=None):
    if isinstance(x, Variable):
        return x
    if isinstance(x, (list, tuple)) and builtins.any(isinstance(e, Variable) for e in x):
        return stack([as_tensor_variable(e) for e in x])
    return Constant(x, name=name)


def constant(x, name=None, ndim=None, dtype=None):
    return Constant(x, dtype=dtype, name=name)


# ------------------------------------------------------------------------------------------------
# evaluation
# ------------------------------------------------------------------------------------------------
def _toposort(outputs, stop):
    order, seen = [], set()
    stack_ = [(o, False) for o in reversed(outputs)]
    while stack_:
        node, done = stack_.pop()
        if done:
            order.append(node)
            continue
        if id(node) in seen or id(node) in stop:
            continue
        seen.add(id(node))
        stack_.append((node, True))
        for i in reversed(node.inputs):
            if id(i) not in seen and id(i) not in stop:
                stack_.append((i, False))
    return order


def ancestors(outputs):
    return _toposort(list(outputs), {})


class _Memo(dict):
    """id(node) -> value, keeping the nodes alive."""


def evaluate(outputs, env, memo=None):
    """Evaluates ``outputs`` given ``env`` {placeholder Variable: torch tensor}.  ``memo`` (id -> tensor) is
    shared within one function call so every node -- in particular every random draw -- is computed once."""
    memo = {"__draws__": [], "__keep__": []} if memo is None else memo
    for k, v in env.items():
        memo[id(k)] = v
    for node in _toposort(list(outputs), memo):
        if id(node) in memo:
            continue
        if isinstance(node, SharedVariable):
            v = node.container[0]
            if node.dtype != "rng" and v.dtype.is_floating_point and not v.requires_grad:
                # one autograd leaf per storage for the duration of a call
                key = ("leaf", id(node.container))
                if key not in memo:
                    memo[key] = v.detach().clone().requires_grad_(True)
                v = memo[key]
            memo[id(node)] = v
        elif isinstance(node, Constant):
            memo[id(node)] = node.value
        elif isinstance(node, TensorPlaceholder):
            raise ValueError("missing input for %r" % node)
        else:
            memo[id(node)] = node.fn(*[memo[id(i)] for i in node.inputs], **getattr(node, "kw", {})) \
                if not getattr(node, "wants_memo", False) else node.fn(memo)
    return [memo[id(o)] for o in outputs]


# ------------------------------------------------------------------------------------------------
# op constructors
# ------------------------------------------------------------------------------------------------
def _true_div(a, b):
    return torch.true_divide(a, b)


def _ew(op, a, b, force_float=False):
    a, b = as_tensor_variable(a), as_tensor_variable(b)
    dt = floatX if force_float else _up(a.dtype, b.dtype)

    def f(x, y):
        if x.dtype != y.dtype:
            if x.dtype.is_floating_point or y.dtype.is_floating_point or force_float:
                x, y = x.to(F64), y.to(F64)
        elif force_float and not x.dtype.is_floating_point:
            x, y = x.to(F64), y.to(F64)
        return op(x, y)
    return Variable(f, [a, b], builtins.max(a.ndim, b.ndim), dt)


def _ew1(op, a, dtype=None):
    a = as_tensor_variable(a)
    return Variable(op, [a], a.ndim, dtype or a.dtype)


def _ewf(op, a):
    """Elementwise op that always yields floats (exp, log, ...)."""
    a = as_tensor_variable(a)
    return Variable(lambda v: op(v.to(F64)), [a], a.ndim, floatX)


def _cmp(op, a, b):
    a, b = as_tensor_variable(a), as_tensor_variable(b)
    return Variable(lambda x, y: op(x, y).to(I64), [a, b], builtins.max(a.ndim, b.ndim), "int8")


def _axes(axis, ndim):
    if axis is None:
        return None
    if isinstance(axis, (int, np.integer)):
        axis = [int(axis)]
    return tuple(int(a) % ndim for a in axis)


def _reduce(kind, x, axis, keepdims=False):
    x = as_tensor_variable(x)
    ax = _axes(axis, x.ndim)
    nd = x.ndim if keepdims else (0 if ax is None else x.ndim - len(ax))
    floaty = kind in ("mean", "var")

    def f(v):
        if floaty:
            v = v.to(F64)
        dims = tuple(range(v.dim())) if ax is None else ax
        if kind == "sum":
            return v.sum() if (ax is None and not keepdims) else v.sum(dim=dims, keepdim=keepdims)
        if kind == "mean":
            return v.mean() if (ax is None and not keepdims) else v.mean(dim=dims, keepdim=keepdims)
        if kind == "var":  # biased, like theano.tensor.var
            m = v.mean(dim=dims, keepdim=True)
            r = ((v - m) ** 2).mean(dim=dims, keepdim=keepdims)
            return r
        if kind in ("max", "min"):
            r = v
            fn = torch.amax if kind == "max" else torch.amin
            return fn(r, dim=dims, keepdim=keepdims)
        raise ValueError(kind)
    return Variable(f, [x], nd, floatX if floaty else x.dtype)


def sum(x, axis=None, keepdims=False): return _reduce("sum", x, axis, keepdims)
def mean(x, axis=None, keepdims=False): return _reduce("mean", x, axis, keepdims)
def var(x, axis=None, keepdims=False): return _reduce("var", x, axis, keepdims)
def max(x, axis=None, keepdims=False): return _reduce("max", x, axis, keepdims)
def min(x, axis=None, keepdims=False): return _reduce("min", x, axis, keepdims)


def argmax(x, axis=None, keepdims=False):
    x = as_tensor_variable(x)
    return Variable(lambda v: torch.argmax(v, dim=axis), [x], x.ndim - 1 if axis is not None else 0, "int64")


def exp(x): return _ewf(torch.exp, x)
def log(x): return _ewf(torch.log, x)
def sqrt(x): return _ewf(torch.sqrt, x)
def tanh(x): return _ewf(torch.tanh, x)
def expm1(x): return _ewf(torch.expm1, x)
def inv(x): return _ewf(torch.reciprocal, x)
def sqr(x): return _ew1(lambda v: v * v, x)
square = sqr
def abs_(x): return _ew1(torch.abs, x)
def neg(x): return _ew1(torch.neg, x)
def add(a, b): return _ew(torch.add, a, b)
def sub(a, b): return _ew(torch.sub, a, b)
def mul(a, b): return _ew(torch.mul, a, b)
def true_div(a, b): return _ew(_true_div, a, b, force_float=True)
def maximum(a, b): return _ew(torch.maximum, a, b)
def minimum(a, b): return _ew(torch.minimum, a, b)
def eq(a, b): return _cmp(torch.eq, a, b)
def neq(a, b): return _cmp(torch.ne, a, b)
def lt(a, b): return _cmp(torch.lt, a, b)
def le(a, b): return _cmp(torch.le, a, b)
def gt(a, b): return _cmp(torch.gt, a, b)
def ge(a, b): return _cmp(torch.ge, a, b)


def clip(x, lo, hi):
    return minimum(maximum(x, lo), hi)


def switch(c, a, b):
    c, a, b = as_tensor_variable(c), as_tensor_variable(a), as_tensor_variable(b)

    def f(cv, av, bv):
        if av.dtype != bv.dtype:
            av, bv = av.to(F64), bv.to(F64)
        return torch.where(cv != 0, av, bv)
    return Variable(f, [c, a, b], builtins.max(c.ndim, a.ndim, b.ndim), _up(a.dtype, b.dtype))


def cast(x, dtype):
    x = as_tensor_variable(x)
    if dtype == "floatX":
        dtype = floatX
    if _is_float(dtype):
        return Variable(lambda v: v.to(F64), [x], x.ndim, dtype)
    return Variable(lambda v: v.to(I64), [x], x.ndim, dtype)


def dimshuffle(x, pattern):
    x = as_tensor_variable(x)
    pattern = tuple(pattern)
    kept = [p for p in pattern if p != "x"]

    def f(v):
        dropped = [d for d in range(v.dim()) if d not in kept]
        for d in dropped:
            if v.shape[d] != 1:
                raise ValueError("dimshuffle drops a non-broadcastable dimension")
        v = v.permute(*(kept + dropped)).reshape([v.shape[k] for k in kept])
        for pos, p in enumerate(pattern):
            if p == "x":
                v = v.unsqueeze(pos)
        return v
    return Variable(f, [x], len(pattern), x.dtype)


def transpose(x, axes=None):
    x = as_tensor_variable(x)
    if axes is None:
        axes = tuple(reversed(range(x.ndim)))
    return Variable(lambda v: v.permute(*axes), [x], x.ndim, x.dtype)


def _shape_inputs(shape):
    """Splits a shape spec (ints and scalar Variables) into (template, symbolic inputs)."""
    if isinstance(shape, Variable):
        return None, [shape]
    shape = list(shape) if isinstance(shape, (list, tuple)) else [shape]
    return shape, [s for s in shape if isinstance(s, Variable)]


def _concrete_shape(template, vals):
    if template is None:
        return tuple(int(s) for s in vals[0])
    it = iter(vals)
    return tuple(int(next(it)) if isinstance(s, Variable) else int(s) for s in template)


def reshape(x, shape, ndim=None):
    x = as_tensor_variable(x)
    template, sym = _shape_inputs(shape)
    nd = ndim if ndim is not None else len(template)
    return Variable(lambda v, *s: v.reshape(_concrete_shape(template, s)), [x] + sym, nd, x.dtype)


def zeros(shape, dtype=None):
    template, sym = _shape_inputs(shape)
    dt = dtype or floatX
    td = F64 if _is_float(dt) else I64
    nd = len(template)
    return Variable(lambda *s: torch.zeros(_concrete_shape(template, s), dtype=td), sym, nd, dt)


def ones(shape, dtype=None):
    template, sym = _shape_inputs(shape)
    dt = dtype or floatX
    td = F64 if _is_float(dt) else I64
    return Variable(lambda *s: torch.ones(_concrete_shape(template, s), dtype=td), sym, len(template), dt)


def zeros_like(x): return _ew1(torch.zeros_like, x)
def ones_like(x): return _ew1(torch.ones_like, x)


def eye(n, m=None, k=0, dtype=None):
    m = n if m is None else m
    n, m = as_tensor_variable(n), as_tensor_variable(m)
    return Variable(lambda a, b: torch.eye(int(a), int(b), dtype=F64), [n, m], 2, dtype or floatX)


def arange(start, stop=None, step=1, dtype=None):
    if stop is None:
        start, stop = 0, start
    a, b, c = as_tensor_variable(start), as_tensor_variable(stop), as_tensor_variable(step)
    dt = dtype or _up(a.dtype, b.dtype, c.dtype)
    td = F64 if _is_float(dt) else I64

    def f(x, y, z):
        x, y, z = (float(x), float(y), float(z)) if td == F64 else (int(x), int(y), int(z))
        return torch.arange(x, y, z, dtype=td)
    return Variable(f, [a, b, c], 1, dt)


def concatenate(tensors, axis=0):
    ts = [as_tensor_variable(t) for t in tensors]
    dt = _up(*[t.dtype for t in ts])

    def f(*vs):
        if builtins.any(v.dtype.is_floating_point for v in vs):
            vs = [v.to(F64) for v in vs]
        return torch.cat(list(vs), dim=axis)
    return Variable(f, ts, ts[0].ndim, dt)


def stack(tensors, axis=0):
    ts = [as_tensor_variable(t) for t in tensors]
    dt = _up(*[t.dtype for t in ts])

    def f(*vs):
        if builtins.any(v.dtype.is_floating_point for v in vs):
            vs = [v.to(F64) for v in vs]
        return torch.stack(list(vs), dim=axis)
    return Variable(f, ts, ts[0].ndim + 1, dt)


def tile(x, reps, ndim=None):
    x = as_tensor_variable(x)
    template, sym = _shape_inputs(reps)
    return Variable(lambda v, *s: v.repeat(*_concrete_shape(template, s)), [x] + sym,
                    builtins.max(x.ndim, len(template)), x.dtype)


def cumsum(x, axis=None):
    x = as_tensor_variable(x)
    if axis is None:
        return Variable(lambda v: torch.cumsum(v.reshape(-1), 0), [x], 1, x.dtype)
    return Variable(lambda v: torch.cumsum(v, axis), [x], x.ndim, x.dtype)


def dot(a, b):
    a, b = as_tensor_variable(a), as_tensor_variable(b)
    nd = a.ndim + b.ndim - 2 if (a.ndim and b.ndim) else builtins.max(a.ndim, b.ndim)
    return Variable(lambda x, y: torch.matmul(x.to(F64), y.to(F64)) if (x.dim() and y.dim()) else x * y, [a, b], nd, floatX)


def tensordot(a, b, axes=2):
    a, b = as_tensor_variable(a), as_tensor_variable(b)
    if isinstance(axes, (int, np.integer)):
        n = int(axes)
        axes = [list(range(a.ndim - n, a.ndim)), list(range(n))]
    ax_a = [int(i) for i in (axes[0] if isinstance(axes[0], (list, tuple)) else [axes[0]])]
    ax_b = [int(i) for i in (axes[1] if isinstance(axes[1], (list, tuple)) else [axes[1]])]
    nd = a.ndim + b.ndim - 2 * len(ax_a)
    return Variable(lambda x, y: torch.tensordot(x.to(F64), y.to(F64), dims=(ax_a, ax_b)), [a, b], nd, floatX)


def batched_tensordot(x, y, axes=2):
    """theano.tensor.batched_tensordot: axis 0 of both operands is the batch; ``axes`` index the FULL tensors
    (batch axis included in the numbering), contracted pairwise; result = (batch, free x axes..., free y axes...)."""
    x, y = as_tensor_variable(x), as_tensor_variable(y)
    if isinstance(axes, (int, np.integer)):
        n = int(axes)
        axes = [list(range(x.ndim - n, x.ndim)), list(range(1, n + 1))]
    ax_x = [int(i) for i in axes[0]]
    ax_y = [int(i) for i in axes[1]]
    if 0 in ax_x or 0 in ax_y:
        raise ValueError("batch axis cannot be contracted")
    letters = "abcdefghijklmnopqrstuvw"
    lx = ["z"] + [letters[i] for i in range(x.ndim - 1)]
    ly = ["z"] + [letters[x.ndim - 1 + i] for i in range(y.ndim - 1)]
    for i, j in zip(ax_x, ax_y):
        ly[j] = lx[i]
    out = ["z"] + [c for k, c in enumerate(lx) if k and k not in ax_x] + [c for k, c in enumerate(ly) if k and k not in ax_y]
    spec = "%s,%s->%s" % ("".join(lx), "".join(ly), "".join(out))
    return Variable(lambda u, v: torch.einsum(spec, u.to(F64), v.to(F64)), [x, y], len(out), floatX)


def _subtensor_write(sub, value, inc):
    if sub._subtensor is None:
        raise TypeError("set/inc_subtensor needs the result of an indexing expression")
    base, idx, sym, build = sub._subtensor
    value = as_tensor_variable(value)

    def f(b, val, *s):
        out = b.clone()
        if val.dtype != out.dtype:
            out, val = out.to(F64), val.to(F64)
        ix = build(s)
        if inc:
            out[ix] = out[ix] + val
        else:
            out[ix] = val
        return out
    return Variable(f, [base, value] + list(sym), base.ndim, _up(base.dtype, value.dtype))


def inc_subtensor(x, y, **kw): return _subtensor_write(x, y, True)
def set_subtensor(x, y, **kw): return _subtensor_write(x, y, False)


def nonzero(x, return_matrix=False):
    raise NotImplementedError("T.nonzero is only referenced from commented-out reference code")


def split(*a, **k):
    raise NotImplementedError("T.split is only used by dead reference layers (SURVEY 2.1)")


# ------------------------------------------------------------------------------------------------
# types / placeholders
# ------------------------------------------------------------------------------------------------
class TensorType(object):
    def __init__(self, dtype, broadcastable):
        self.dtype = floatX if dtype == "floatX" else dtype
        self.broadcastable = tuple(broadcastable)
        self.ndim = len(self.broadcastable)

    def __call__(self, name=None):
        return TensorPlaceholder(self.ndim, self.dtype, name)


def _placeholder_factory(ndim, default_dtype):
    def make(name=None, dtype=None):
        return TensorPlaceholder(ndim, dtype or default_dtype, name)
    return make


scalar = _placeholder_factory(0, floatX)
vector = _placeholder_factory(1, floatX)
matrix = _placeholder_factory(2, floatX)
tensor3 = _placeholder_factory(3, floatX)
tensor4 = _placeholder_factory(4, floatX)
fscalar, fvector, fmatrix, ftensor3, ftensor4 = scalar, vector, matrix, tensor3, tensor4
iscalar = _placeholder_factory(0, "int32")
ivector = _placeholder_factory(1, "int32")
imatrix = _placeholder_factory(2, "int32")


# ------------------------------------------------------------------------------------------------
# gradients
# ------------------------------------------------------------------------------------------------
class _GradGroup(object):
    """One ``T.grad(cost, wrt_list, consider_constant)`` call: all gradients come from one autograd pass."""

    def __init__(self, cost, wrt, consider_constant):
        self.cost, self.wrt, self.cc = cost, list(wrt), list(consider_constant or [])

    def compute(self, memo):
        key = ("grad", id(self))
        if key in memo:
            return memo[key]
        for w in self.wrt:  # wrt values must exist in the main pass (leaves or inputs)
            evaluate([w], {}, memo)
        if self.cc:
            # Re-evaluate the cost with the constants cut: leaves, inputs and random draws are reused from the
            # main pass, the consider_constant nodes are replaced by their detached main-pass values.
            evaluate(self.cc, {}, memo)
            sub = {}
            for k, v in memo.items():
                if not isinstance(k, int):  # autograd leaves, cached gradient groups, bookkeeping lists
                    sub[k] = v
            for nid in memo["__keep__"]:  # inputs, givens and random draws are evaluated once per call
                sub[nid] = memo[nid]
            for w in self.wrt:
                sub[id(w)] = memo[id(w)]
            for c in self.cc:
                sub[id(c)] = memo[id(c)].detach()
            cost = evaluate([self.cost], {}, sub)[0]
        else:
            cost = evaluate([self.cost], {}, memo)[0]
        wv = [memo[id(w)] for w in self.wrt]
        for w, v in zip(self.wrt, wv):
            if not v.requires_grad:
                raise ValueError("cannot differentiate with respect to %r" % w)
        gs = torch.autograd.grad(cost, wv, retain_graph=True, allow_unused=True)
        gs = [torch.zeros_like(v) if g is None else g for g, v in zip(gs, wv)]
        memo[key] = gs
        return gs


def grad(cost, wrt, consider_constant=None, known_grads=None, disconnected_inputs="raise", **kw):
    single = not isinstance(wrt, (list, tuple))
    wl = [wrt] if single else list(wrt)
    grp = _GradGroup(cost, wl, consider_constant)
    outs = []
    for i, w in enumerate(wl):
        v = Variable(None, [cost] + wl, w.ndim, w.dtype, name="grad(%s)" % w.name)
        v.wants_memo = True
        v.fn = (lambda memo, i=i: grp.compute(memo)[i])
        v.grad_group = grp
        outs.append(v)
    return outs[0] if single else outs


# ------------------------------------------------------------------------------------------------
# theano.function / theano.clone
# ------------------------------------------------------------------------------------------------
class Function(object):
    """theano.function(inputs, outputs, updates=, givens=).  Updates are computed from the pre-call state and
    then assigned; ``default_update`` attributes of shared variables reachable from the outputs/updates are
    honoured (BatchNormLayer's running averages); when several aliases of one storage carry a default update
    (one per get_output pass) the last one in graph order wins -- Theano leaves that order unspecified too."""

    def __init__(self, inputs, outputs=None, updates=None, givens=None, name=None, on_unused_input=None, **kw):
        self.inputs = list(inputs)
        self.single = not isinstance(outputs, (list, tuple))
        self.outputs = [] if outputs is None else ([outputs] if self.single else list(outputs))
        if updates is None:
            updates = []
        self.updates = OrderedDict(updates.items() if isinstance(updates, dict) else updates)
        self.givens = list(givens.items()) if isinstance(givens, dict) else list(givens or [])
        self.name = name
        self.last_draws = []

    def __call__(self, *args):
        if len(args) != len(self.inputs):
            raise TypeError("%s: expected %d inputs, got %d" % (self.name, len(self.inputs), len(args)))
        memo = {"__draws__": [], "__keep__": [], "__nodes__": list(self.inputs)}
        for var, val in zip(self.inputs, args):
            t = _to_torch(np.asarray(val), var.dtype)
            if t.dtype.is_floating_point:
                t = t.clone().requires_grad_(True)
            memo[id(var)] = t
            memo["__keep__"].append(id(var))
        for var, expr in self.givens:
            expr = as_tensor_variable(expr)
            memo[id(var)] = evaluate([expr], {}, memo)[0]
            memo["__keep__"].append(id(var))
        targets = list(self.outputs) + [as_tensor_variable(e) for e in self.updates.values()]
        # default updates of shared variables in the graph
        dflt = []
        for node in ancestors(targets):
            if isinstance(node, SharedVariable) and getattr(node, "default_update", None) is not None \
                    and node not in self.updates:
                dflt.append((node, as_tensor_variable(node.default_update)))
        with torch.enable_grad():
            vals = evaluate(targets + [e for _, e in dflt], {}, memo)
        n_out, n_up = len(self.outputs), len(self.updates)
        outs = [_to_numpy(v) for v in vals[:n_out]]
        new = [v.detach() for v in vals[n_out:]]
        for (sv, _), v in zip(list(self.updates.items()) + dflt, new):
            if sv.dtype == "rng":
                sv.container[0] = v
            else:
                sv.container[0] = v.to(F64) if _is_float(sv.dtype) else v.to(I64)
        self.last_draws = memo["__draws__"]
        if self.single:
            return outs[0] if outs else None
        return outs


def function(inputs, outputs=None, updates=None, givens=None, **kw):
    return Function(inputs, outputs, updates, givens, **kw)


def clone(output, replace=None, strict=True, share_inputs=True, copy_inputs=None):
    if isinstance(output, SharedVariable) and not replace:
        return output if share_inputs else output.clone()
    raise NotImplementedError("theano.clone is only needed for the BatchNormLayer running-average aliases")


# ------------------------------------------------------------------------------------------------
# random streams
# ------------------------------------------------------------------------------------------------
class RandomStreams(object):
    """theano.tensor.shared_randomstreams.RandomStreams [recalled]: ``gen_seedgen = RandomState(seed)``; every
    random variable gets its own ``RandomState(int(gen_seedgen.randint(2**30)))`` held in a shared variable;
    ``seed(s)`` reseeds gen_seedgen and re-creates every stream from it in creation order."""

    def __init__(self, seed=None):
        self.default_instance_seed = seed
        self.gen_seedgen = np.random.RandomState(seed)
        self.state_updates = []

    def seed(self, seed=None):
        if seed is None:
            seed = self.default_instance_seed
        self.gen_seedgen.seed(seed)
        for old_r, _ in self.state_updates:
            old_r_seed = self.gen_seedgen.randint(2 ** 30)
            old_r.set_value(np.random.RandomState(int(old_r_seed)), borrow=True)

    def _new_stream(self):
        seed = int(self.gen_seedgen.randint(2 ** 30))
        sv = SharedVariable(np.random.RandomState(seed), name="rng")
        sv.child_seed = seed
        self.state_updates.append([sv, sv])
        return sv

    def permutation(self, size=None, n=1, ndim=None):
        raise NotImplementedError

    def shuffle_row_elements(self, input):
        """perm = permutation(size=input.shape[:-1], n=input.shape[-1]); out[..., i] = input[..., perm[..., i]]
        (theano.tensor.raw_random.permutation_helper + permute_row_elements).  The RandomState advances in
        place when the node is evaluated, as Theano's in-place random state update does."""
        input = as_tensor_variable(input)
        sv = self._new_stream()

        def f(v, rs):
            lead = tuple(v.shape[:-1])
            n = int(v.shape[-1])
            perm = np.empty(lead + (n,), dtype=np.int64)
            for i in np.ndindex(*lead):
                perm[i] = rs.permutation(n)
            return torch.gather(v, v.dim() - 1, torch.from_numpy(perm))
        return Variable(f, [input, sv], input.ndim, input.dtype)


class MRG_RandomStreams(object):
    """theano.sandbox.rng_mrg.MRG_RandomStreams -- the MRG31k3p generator is NOT restated (module docstring)."""

    def __init__(self, seed=12345, use_cuda=None):
        self.seed_value = seed
        self.rstate = np.asarray([seed] * 6, dtype="int32") if isinstance(seed, (int, np.integer)) else np.asarray(seed)
        self.state_updates = []
        self._np = np.random.RandomState(int(np.asarray(seed).ravel()[0]) % (2 ** 31))

    def _shape(self, size):
        return _shape_inputs(size)

    def normal(self, size, avg=0.0, std=1.0, ndim=None, dtype=None, nstreams=None):
        template, sym = _shape_inputs(size)
        v = Variable(None, sym, len(template), dtype or floatX, name="mrg_normal")
        v.wants_memo = True
        rs = self._np

        def f(memo):
            shp = _concrete_shape(template, [memo[id(s)] for s in sym])
            draw = rs.standard_normal(shp).astype(np.float32)  # float32 like the reference's floatX stream
            memo["__draws__"].append(("normal", draw))
            memo["__keep__"].append(id(v))
            return torch.from_numpy(draw.astype(np.float64)) * std + avg
        v.fn = f
        return v

    def multinomial(self, size=None, n=1, pvals=None, ndim=None, dtype="int64", nstreams=None):
        pvals = as_tensor_variable(pvals)

        def f(p):
            onehot = ((p == 0) | (p == 1)).all() and bool((p.sum(dim=1) == 1).all())
            if not onehot:
                raise NotImplementedError("MRG31k3p multinomial is only restated for one-hot probability rows "
                                          "(mask_generator.py:75-91 with mask_distribution l = 0)")
            return p.clone()
        return Variable(f, [pvals], 2, dtype)
