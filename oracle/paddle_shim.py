"""TEST INFRASTRUCTURE (oracle/): a torch-CPU stand-in for the `paddle` package, just large enough to EXECUTE THE REFERENCE'S OWN
PYTHON for the hot path and its callers -- ppdiffusers/ppdiffusers/models/*.py (UNet, ControlNet, DiT, SD3 MMDiT, AutoencoderKL and
their blocks, attention processors, embeddings, LoRA layers), schedulers/scheduling_*.py, transformers/{clip,t5}/modeling.py,
image_processor.py and pipelines/*/pipeline_*.py -- loaded unmodified from /root/reference by oracle/reference_runner.py, in this
container, where PaddlePaddle itself cannot be installed.

Why: the oracle (oracle/unet_ref.py ...) is a restatement of that code; until round 3 its whole-model numerics were "unpinned"
(only the RNG-free scheduler / embedding vectors of the reference's tests pinned it). Running the reference's module graph --
its constructors, its forward methods, its attention processors, its reshapes / transposes / concatenations, its sampling and
pipeline loops, verbatim -- on the oracle's parameters and inputs and comparing the outputs pins the STRUCTURE of the restatement
(what is wired to what, which axis, which scale, which epsilon, which order of random draws) to the reference itself. What it does
not pin is Paddle's own kernels: every array operation below is torch's fp32 CPU implementation of the documented Paddle
semantics (paddle.nn.Linear keeps weight [in, out]; Tensor.transpose takes a permutation; reshape's 0 copies a dimension; chunk /
split / concat take `axis`; a tied parameter is listed once in a state dict; GroupNorm / LayerNorm take `epsilon`).

Only what those files use is implemented; anything else raises AttributeError / NotImplementedError loudly. Nothing outside
tests/, scripts/make_reference_golden.py and scripts/check_parity_fixtures_against_reference.py imports this module.
"""
from __future__ import annotations

import contextlib
import math
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as TF

# ------------------------------------------------------------------------------------------------------------------ dtypes
float32, float16, bfloat16, float64 = torch.float32, torch.float16, torch.bfloat16, torch.float64
int64, int32, int8, uint8, bool_ = torch.int64, torch.int32, torch.int8, torch.uint8, torch.bool
_DT = {"float32": float32, "float16": float16, "bfloat16": bfloat16, "float64": float64, "int64": int64, "int32": int32, "bool": bool_,
       "uint8": uint8, "int8": int8}


def _dtype(d):
    if d is None or isinstance(d, torch.dtype):
        return d
    if isinstance(d, str):
        return _DT[d.replace("paddle.", "")]
    if isinstance(d, np.dtype) or (isinstance(d, type) and issubclass(d, np.generic)):
        return torch.from_numpy(np.zeros(1, dtype=d)).dtype
    raise TypeError(f"paddle_shim: dtype {d!r}")


# ------------------------------------------------------------------------------------------------------------------ Tensor
def _u(x):
    """unwrap"""
    if isinstance(x, Tensor):
        return x.t
    if isinstance(x, (list, tuple)):
        return type(x)(_u(v) for v in x)
    return x


def _w(t):
    return Tensor(t) if isinstance(t, torch.Tensor) else t


def _shape(args):
    """reshape([a, b]) | reshape(a, b) | entries may be 0-d Tensors"""
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        args = args[0]
    return [int(_u(a)) for a in args]


class Tensor:
    """paddle.Tensor semantics on a torch tensor `t` (fp32 CPU math)."""

    def __init__(self, t, stop_gradient=True):
        if isinstance(t, Tensor):
            t = t.t
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        self.t = t
        self.stop_gradient = stop_gradient
        self.name = "shim_tensor"

    # -- metadata
    @property
    def shape(self):
        return list(self.t.shape)

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def ndim(self):
        return self.t.dim()

    @property
    def size(self):
        return self.t.numel()

    @property
    def place(self):
        return "cpu"

    @property
    def T(self):
        return Tensor(self.t.T)

    def dim(self):
        return self.t.dim()

    def numel(self):
        return Tensor(torch.tensor(self.t.numel()))

    def __len__(self):
        return self.t.shape[0]

    def __repr__(self):
        return f"shim.Tensor({self.t!r})"

    def __bool__(self):
        return bool(self.t)

    def __int__(self):
        return int(self.t)

    def __float__(self):
        return float(self.t)

    def __index__(self):
        return int(self.t)

    def __array__(self, dtype=None, copy=None):
        a = self.t.detach().numpy()
        return a.astype(dtype) if dtype is not None else a

    def item(self, *a):
        return self.t.item()

    def numpy(self):
        return self.t.detach().numpy()

    def tolist(self):
        return self.t.tolist()

    def is_floating_point(self):
        return self.t.is_floating_point()

    # -- conversions / copies
    def cast(self, dtype):
        return Tensor(self.t.to(_dtype(dtype)))

    astype = cast

    def to(self, *a, **k):   # paddle_patch-style Tensor.to(dtype=...)
        dt = k.get("dtype")
        for v in a:
            if isinstance(v, (torch.dtype, str)) and v not in ("cpu", "gpu"):
                dt = v
        return self.cast(dt) if dt is not None else self

    _to = to   # Paddle's own (private) Tensor._to(dtype=...)

    def clone(self):
        return Tensor(self.t.clone())

    def detach(self):
        return Tensor(self.t.detach())

    def contiguous(self):
        return Tensor(self.t.contiguous())

    def cpu(self):
        return self

    def cuda(self, *a, **k):
        return self

    def set_value(self, v):
        v = _u(v)
        v = torch.as_tensor(v)
        assert tuple(v.shape) == tuple(self.t.shape), (tuple(v.shape), tuple(self.t.shape))
        self.t = v.to(self.t.dtype).clone()

    def copy_(self, other, blocking=True):
        self.set_value(other)
        return self

    def zero_(self):
        self.t.zero_()
        return self

    def fill_(self, v):
        self.t.fill_(v)
        return self

    # -- shape ops
    def reshape(self, *shape, **kw):
        return reshape(self, _shape(shape) if shape else kw["shape"])

    def transpose(self, perm, *rest):
        if rest:   # never the torch form here: paddle's takes ONE permutation
            raise TypeError("paddle_shim: Tensor.transpose(perm) takes a permutation list")
        return Tensor(self.t.permute(*[int(p) for p in perm]))

    def unsqueeze(self, axis):
        t = self.t
        for a in ([axis] if isinstance(axis, int) else list(axis)):
            t = t.unsqueeze(a)
        return Tensor(t)

    def squeeze(self, axis=None):
        if axis is None:
            return Tensor(self.t.squeeze())
        t = self.t
        for a in sorted([axis] if isinstance(axis, int) else list(axis), reverse=True):
            t = t.squeeze(a)
        return Tensor(t)

    def flatten(self, start_axis=0, stop_axis=-1):
        return Tensor(self.t.flatten(start_axis, stop_axis))

    def expand(self, *shape):
        return Tensor(self.t.expand(*_shape(shape)))

    def broadcast_to(self, shape):
        return Tensor(self.t.expand(*_shape((shape,))))

    def expand_as(self, y):
        return Tensor(self.t.expand_as(_u(y)))

    def tile(self, repeat_times):
        return Tensor(self.t.repeat(*_shape((repeat_times,))))

    def repeat_interleave(self, repeats, axis=None):
        return Tensor(self.t.repeat_interleave(_u(repeats), dim=axis))

    def chunk(self, chunks, axis=0):
        return chunk(self, chunks, axis)

    def split(self, num_or_sections, axis=0):
        return split(self, num_or_sections, axis)

    def flip(self, axis):
        return flip(self, axis)

    def unbind(self, axis=0):
        return [Tensor(v) for v in self.t.unbind(axis)]

    def moveaxis(self, s, d):
        return Tensor(self.t.movedim(s, d))

    # -- math
    def matmul(self, y, transpose_x=False, transpose_y=False):
        return matmul(self, y, transpose_x, transpose_y)

    def sum(self, axis=None, dtype=None, keepdim=False):
        return sum(self, axis, dtype, keepdim)

    def mean(self, axis=None, keepdim=False):
        return mean(self, axis, keepdim)

    def std(self, axis=None, unbiased=True, keepdim=False):
        return Tensor(self.t.std(unbiased=unbiased) if axis is None else self.t.std(dim=axis, unbiased=unbiased, keepdim=keepdim))

    def var(self, axis=None, unbiased=True, keepdim=False):
        return Tensor(self.t.var(unbiased=unbiased) if axis is None else self.t.var(dim=axis, unbiased=unbiased, keepdim=keepdim))

    def max(self, axis=None, keepdim=False):
        return Tensor(self.t.max()) if axis is None else Tensor(self.t.amax(dim=axis, keepdim=keepdim))

    def min(self, axis=None, keepdim=False):
        return Tensor(self.t.min()) if axis is None else Tensor(self.t.amin(dim=axis, keepdim=keepdim))

    def norm(self, p=2, axis=None, keepdim=False):
        return Tensor(torch.linalg.vector_norm(self.t, ord=p, dim=axis, keepdim=keepdim))

    def softmax(self, axis=-1):
        return Tensor(torch.softmax(self.t, dim=axis))

    def sqrt(self):
        return Tensor(self.t.sqrt())

    def rsqrt(self):
        return Tensor(self.t.rsqrt())

    def exp(self):
        return Tensor(self.t.exp())

    def log(self):
        return Tensor(self.t.log())

    def abs(self):
        return Tensor(self.t.abs())

    def pow(self, y):
        return Tensor(self.t.pow(_u(y)))

    def clip(self, min=None, max=None):
        return Tensor(self.t.clamp(_u(min), _u(max)))

    def scale(self, scale=1.0, bias=0.0):
        return Tensor(self.t * scale + bias)

    def masked_fill(self, mask, value):
        return Tensor(self.t.masked_fill(_u(mask), value))

    def isnan(self):
        return Tensor(self.t.isnan())

    def nonzero(self, as_tuple=False):
        return nonzero(self, as_tuple)

    def gather_nd(self, index):
        return gather_nd(self, index)

    def argmax(self, axis=None, keepdim=False, dtype="int64"):
        return argmax(self, axis, keepdim, dtype)

    def cumprod(self, dim=None):
        return cumprod(self, dim)

    def round(self):
        return Tensor(self.t.round())

    def sin(self):
        return Tensor(self.t.sin())

    def cos(self):
        return Tensor(self.t.cos())

    def atan(self):
        return Tensor(self.t.atan())

    def any(self):
        return Tensor(self.t.any())

    def all(self):
        return Tensor(self.t.all())

    def float(self):
        return Tensor(self.t.float())

    # -- indexing
    def __getitem__(self, idx):
        return Tensor(self.t[_u(idx) if not isinstance(idx, tuple) else tuple(_u(i) for i in idx)])

    def __setitem__(self, idx, v):
        self.t[_u(idx) if not isinstance(idx, tuple) else tuple(_u(i) for i in idx)] = _u(v)

    def __iter__(self):
        return (Tensor(v) for v in self.t)


def _binop(name, rname=None):
    def f(self, other):
        return Tensor(getattr(self.t, name)(_u(other)))
    setattr(Tensor, name, f)
    if rname:
        def r(self, other):
            o = _u(other)
            o = o if isinstance(o, torch.Tensor) else torch.as_tensor(o, dtype=self.t.dtype if isinstance(o, float) or self.t.is_floating_point() else None)
            return Tensor(getattr(o, name)(self.t))
        setattr(Tensor, rname, r)


for _n, _r in (("__add__", "__radd__"), ("__sub__", "__rsub__"), ("__mul__", "__rmul__"), ("__truediv__", "__rtruediv__"),
               ("__floordiv__", "__rfloordiv__"), ("__pow__", "__rpow__"), ("__matmul__", "__rmatmul__"), ("__mod__", None),
               ("__eq__", None), ("__ne__", None), ("__lt__", None), ("__le__", None), ("__gt__", None), ("__ge__", None),
               ("__and__", None), ("__or__", None), ("__xor__", None)):
    _binop(_n, _r)
Tensor.__neg__ = lambda self: Tensor(-self.t)
Tensor.__invert__ = lambda self: Tensor(~self.t)
Tensor.__hash__ = lambda self: id(self)
Tensor.__iadd__ = lambda self, o: Tensor(self.t + _u(o))
Tensor.__imul__ = lambda self, o: Tensor(self.t * _u(o))


class Parameter(Tensor):
    def __init__(self, t, trainable=True):
        super().__init__(t, stop_gradient=not trainable)
        self.trainable = trainable


# ------------------------------------------------------------------------------------------------------------------ functions
def to_tensor(data, dtype=None, place=None, stop_gradient=True):
    if isinstance(data, Tensor):
        t = data.t
    elif isinstance(data, (list, tuple)) and any(isinstance(v, Tensor) for v in data):
        t = torch.stack([torch.as_tensor(_u(v)) for v in data])
    else:
        t = torch.as_tensor(np.asarray(data) if isinstance(data, np.ndarray) else data)
        if t.dtype == torch.float64 and not isinstance(data, np.ndarray):
            t = t.float()   # python floats become the default dtype
    return Tensor(t.to(_dtype(dtype)) if dtype is not None else t)


def concat(x, axis=0, name=None):
    return Tensor(torch.cat([_u(v) for v in x], dim=int(_u(axis))))


def stack(x, axis=0, name=None):
    return Tensor(torch.stack([_u(v) for v in x], dim=axis))


def matmul(x, y, transpose_x=False, transpose_y=False, name=None):
    a, b = _u(x), _u(y)
    if transpose_x:
        a = a.transpose(-1, -2)
    if transpose_y:
        b = b.transpose(-1, -2)
    return Tensor(torch.matmul(a, b))


def bmm(x, y):
    return Tensor(torch.bmm(_u(x), _u(y)))


def einsum(eq, *ops):
    return Tensor(torch.einsum(eq, *[_u(o) for o in ops]))


def chunk(x, chunks, axis=0, name=None):
    t = _u(x)
    assert t.shape[axis] % chunks == 0, "paddle.chunk: the axis must divide evenly"
    return [Tensor(v) for v in t.chunk(chunks, dim=axis)]


def split(x, num_or_sections, axis=0, name=None):
    t = _u(x)
    if isinstance(num_or_sections, int):
        assert t.shape[axis] % num_or_sections == 0
        return [Tensor(v) for v in t.split(t.shape[axis] // num_or_sections, dim=axis)]
    secs = [int(_u(s)) for s in num_or_sections]
    if -1 in secs:
        known = builtins_sum(s for s in secs if s != -1)
        secs[secs.index(-1)] = t.shape[axis] - known
    return [Tensor(v) for v in t.split(secs, dim=axis)]


import builtins as _b  # noqa: E402

builtins_sum = _b.sum


def flip(x, axis, name=None):
    return Tensor(torch.flip(_u(x), dims=[axis] if isinstance(axis, int) else list(axis)))


def zeros(shape, dtype=None):
    return Tensor(torch.zeros(_shape((shape,)), dtype=_dtype(dtype) or float32))


def ones(shape, dtype=None):
    return Tensor(torch.ones(_shape((shape,)), dtype=_dtype(dtype) or float32))


def full(shape, fill_value, dtype=None):
    return Tensor(torch.full(_shape((shape,)), _u(fill_value), dtype=_dtype(dtype) or float32))


def zeros_like(x, dtype=None):
    return Tensor(torch.zeros_like(_u(x), dtype=_dtype(dtype)))


def ones_like(x, dtype=None):
    return Tensor(torch.ones_like(_u(x), dtype=_dtype(dtype)))


def arange(start=0, end=None, step=1, dtype=None):
    if end is None:
        start, end = 0, start
    vals = [_u(start), _u(end), _u(step)]
    if dtype is None:
        dtype = float32 if any(isinstance(v, float) for v in vals) else int64
    return Tensor(torch.arange(vals[0], vals[1], vals[2], dtype=_dtype(dtype)))


def logspace(start, stop, num, base=10.0, dtype=None):
    return Tensor(torch.logspace(_u(start), _u(stop), int(_u(num)), base=base, dtype=_dtype(dtype) or float32))


def linspace(start, stop, num, dtype=None):
    return Tensor(torch.linspace(_u(start), _u(stop), int(_u(num)), dtype=_dtype(dtype) or float32))


def randn(shape, dtype=None):
    """constructors draw initial values with it (Parameter(paddle.randn(...))): placeholders -- load_params then sets every
    parameter by name, and random inputs always come from the caller's generator -- so the shim answers zeros"""
    return zeros(shape, dtype)


rand = randn


def cast(x, dtype):
    return Tensor(_u(x).to(_dtype(dtype)))


def reshape(x, shape):
    t = _u(x)
    shape = _shape((shape,))
    shape = [t.shape[i] if s == 0 else s for i, s in enumerate(shape)]   # paddle: 0 copies the input's dimension at that position
    return Tensor(t.reshape(shape))


def transpose(x, perm):
    return Tensor(_u(x).permute(*perm))


def unsqueeze(x, axis):
    return Tensor(x).unsqueeze(axis)


def squeeze(x, axis=None):
    return Tensor(x).squeeze(axis)


def tile(x, repeat_times):
    return Tensor(x).tile(repeat_times)


def expand(x, shape):
    return Tensor(x).expand(shape)


def sum(x, axis=None, dtype=None, keepdim=False):  # noqa: A001
    t = _u(x)
    return Tensor(t.sum(dtype=_dtype(dtype)) if axis is None else t.sum(dim=axis, keepdim=keepdim, dtype=_dtype(dtype)))


def mean(x, axis=None, keepdim=False):
    t = _u(x)
    return Tensor(t.mean() if axis is None else t.mean(dim=axis, keepdim=keepdim))


def _unary(fn):
    return lambda x, name=None: Tensor(fn(_u(x)))


sin, cos, exp, log, sqrt, rsqrt, tanh, isnan, abs, sigmoid = (_unary(f) for f in (torch.sin, torch.cos, torch.exp, torch.log, torch.sqrt, torch.rsqrt,  # noqa: A001
                                                                                  torch.tanh, torch.isnan, torch.abs, torch.sigmoid))


def triu(x, diagonal=0):
    return Tensor(torch.triu(_u(x), diagonal=diagonal))


def tril(x, diagonal=0):
    return Tensor(torch.tril(_u(x), diagonal=diagonal))


def full_like(x, fill_value, dtype=None):
    return Tensor(torch.full_like(_u(x), _u(fill_value), dtype=_dtype(dtype)))


def masked_fill(x, mask, value):
    return Tensor(_u(x).masked_fill(_u(mask), _u(value)))


def gather_nd(x, index):
    """paddle.gather_nd: index [..., k] addresses the first k axes of x"""
    idx = _u(index).long()
    return Tensor(_u(x)[tuple(idx[..., i] for i in range(idx.shape[-1]))])


def argmax(x, axis=None, keepdim=False, dtype="int64"):
    return Tensor(torch.argmax(_u(x), dim=axis, keepdim=keepdim).to(_dtype(dtype)))


def cumprod(x, dim=None, dtype=None):
    return Tensor(torch.cumprod(_u(x), dim=dim, dtype=_dtype(dtype)))


def cumsum(x, axis=None, dtype=None):
    t = _u(x)
    return Tensor(torch.cumsum(t.flatten() if axis is None else t, dim=0 if axis is None else axis, dtype=_dtype(dtype)))


def isinf(x):
    return Tensor(torch.isinf(_u(x)))


def quantile(x, q, axis=None, keepdim=False):
    return Tensor(torch.quantile(_u(x), q, dim=axis, keepdim=keepdim))


def searchsorted(sorted_sequence, values, out_int32=False, right=False):
    return Tensor(torch.searchsorted(_u(sorted_sequence), _u(values), out_int32=out_int32, right=right))


def nonzero(x, as_tuple=False):
    return Tensor(torch.nonzero(_u(x))) if not as_tuple else tuple(Tensor(v) for v in torch.nonzero(_u(x), as_tuple=True))


def outer(x, y):
    return Tensor(torch.outer(_u(x), _u(y)))


def where(cond, x=None, y=None):
    return Tensor(torch.where(_u(cond), _u(x) if isinstance(_u(x), torch.Tensor) else torch.tensor(_u(x)), _u(y) if isinstance(_u(y), torch.Tensor) else torch.tensor(_u(y))))


def maximum(x, y):
    return Tensor(torch.maximum(_u(x), _u(y)))


def minimum(x, y):
    return Tensor(torch.minimum(_u(x), _u(y)))


def clip(x, min=None, max=None):  # noqa: A002
    return Tensor(_u(x).clamp(_u(min), _u(max)))


def pow(x, y):  # noqa: A001
    return Tensor(torch.pow(_u(x), _u(y)))


def is_tensor(x):
    return isinstance(x, Tensor)


def shape(x):
    return Tensor(torch.tensor(list(_u(x).shape), dtype=torch.int32))


def assign(x, output=None):
    if output is not None:
        output.set_value(x)
        return output
    return Tensor(_u(x).clone())


def get_default_dtype():
    return "float32"


def set_default_dtype(d):
    assert _dtype(d) == float32


def finfo(dtype):
    return torch.finfo(_dtype(dtype))


def iinfo(dtype):
    return torch.iinfo(_dtype(dtype))


def in_dynamic_mode():
    return True


class no_grad(contextlib.ContextDecorator):
    def __enter__(self):
        self._g = torch.no_grad()
        self._g.__enter__()

    def __exit__(self, *a):
        self._g.__exit__(*a)


@contextlib.contextmanager
def dtype_guard(dtype="float32"):
    yield


def create_parameter(shape, dtype=None, attr=None, is_bias=False, default_initializer=None):
    return Parameter(torch.zeros(_shape((shape,)), dtype=_dtype(dtype) or float32))


class Generator:   # annotations only: random draws never come from the shim
    pass


class ParamAttr:
    def __init__(self, *a, **k):
        self.args, self.kw = a, k


# ------------------------------------------------------------------------------------------------------------------ nn
class Layer:
    def __init__(self, name_scope=None, dtype="float32"):
        d = self.__dict__
        d["_sub_layers"] = OrderedDict()
        d["_parameters"] = OrderedDict()
        d["_buffers"] = OrderedDict()
        d["training"] = True
        d["_dtype"] = "float32"

    def __setattr__(self, k, v):
        d = self.__dict__
        if "_sub_layers" not in d:
            raise RuntimeError("paddle_shim: call super().__init__() first")
        for reg in (d["_sub_layers"], d["_parameters"], d["_buffers"]):
            reg.pop(k, None)
        if isinstance(v, Parameter):
            d["_parameters"][k] = v
        elif isinstance(v, Layer):
            d["_sub_layers"][k] = v
        object.__setattr__(self, k, v)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, *a, **k):
        raise NotImplementedError

    def add_sublayer(self, name, layer):
        setattr(self, str(name), layer)
        return layer

    def add_parameter(self, name, p):
        setattr(self, name, p)
        return p

    def register_buffer(self, name, tensor, persistable=True):
        object.__setattr__(self, name, tensor)
        if tensor is not None:
            self.__dict__["_buffers"][name] = (tensor, persistable)

    def create_parameter(self, shape, attr=None, dtype=None, is_bias=False, default_initializer=None):
        return create_parameter(shape, dtype)

    def children(self):
        return iter(self._sub_layers.values())

    def named_children(self):
        return iter(self._sub_layers.items())

    def named_sublayers(self, prefix="", include_self=False, layers_set=None):
        if include_self:
            yield prefix, self
        for n, l in self._sub_layers.items():
            if l is None:
                continue
            p = prefix + ("." if prefix else "") + n
            yield p, l
            yield from l.named_sublayers(prefix=p)

    def sublayers(self, include_self=False):
        return [l for _, l in self.named_sublayers(include_self=include_self)]

    def named_parameters(self, prefix="", include_sublayers=True, _seen=None):
        seen = set() if _seen is None else _seen     # a tied parameter is listed once, under its first name (as Paddle does)
        for n, p in self._parameters.items():
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                yield prefix + ("." if prefix else "") + n, p
        if include_sublayers:
            for n, l in self._sub_layers.items():
                if l is not None:
                    yield from l.named_parameters(prefix + ("." if prefix else "") + n, True, seen)

    def parameters(self, include_sublayers=True):
        return [p for _, p in self.named_parameters(include_sublayers=include_sublayers)]

    def state_dict(self, *a, **k):
        sd = OrderedDict(self.named_parameters())
        for pfx, l in [("", self)] + list(self.named_sublayers()):
            for n, (t, persist) in l._buffers.items():
                if persist:
                    sd[pfx + ("." if pfx else "") + n] = t
        return sd

    def set_state_dict(self, sd, use_structured_name=True):
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        for k, v in sd.items():
            if k in own:
                own[k].set_value(v)
        return missing, unexpected

    set_dict = load_dict = set_state_dict

    def eval(self):
        for l in self.sublayers(include_self=True):
            l.__dict__["training"] = False
        return self

    def train(self):
        for l in self.sublayers(include_self=True):
            l.__dict__["training"] = True
        return self

    def apply(self, fn):
        for l in self.sublayers(include_self=True):
            fn(l)
        return self

    def to(self, *a, **k):
        return self

    def full_name(self):
        return type(self).__name__.lower()

    def extra_repr(self):
        return ""


class LayerList(Layer):
    def __init__(self, sublayers=None):
        super().__init__()
        if sublayers is not None:
            for l in sublayers:
                self.append(l)

    def append(self, l):
        self.add_sublayer(str(len(self._sub_layers)), l)
        return self

    def extend(self, ls):
        for l in ls:
            self.append(l)
        return self

    def insert(self, index, l):
        items = list(self._sub_layers.values())
        items.insert(index, l)
        self._rebuild(items)

    def _rebuild(self, items):
        for k in list(self._sub_layers):
            object.__delattr__(self, k)
        self.__dict__["_sub_layers"].clear()
        for l in items:
            self.append(l)

    def __len__(self):
        return len(self._sub_layers)

    def __iter__(self):
        return iter(self._sub_layers.values())

    def __getitem__(self, i):
        items = list(self._sub_layers.values())
        return LayerList(items[i]) if isinstance(i, slice) else items[i]

    def __setitem__(self, i, l):
        items = list(self._sub_layers.values())
        items[i] = l
        self._rebuild(items)


class Sequential(Layer):
    def __init__(self, *layers):
        super().__init__()
        if len(layers) == 1 and isinstance(layers[0], (list, tuple)) and not isinstance(layers[0], Layer):
            layers = layers[0]
        for i, l in enumerate(layers):
            if isinstance(l, (list, tuple)):
                self.add_sublayer(l[0], l[1])
            else:
                self.add_sublayer(str(i), l)

    def forward(self, x):
        for l in self._sub_layers.values():
            x = l(x)
        return x

    def __getitem__(self, i):
        return list(self._sub_layers.values())[i]

    def __len__(self):
        return len(self._sub_layers)

    def __iter__(self):
        return iter(self._sub_layers.values())


def _has(attr):
    return attr is not False


class Linear(Layer):
    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = Parameter(torch.zeros(in_features, out_features))     # paddle layout: [in, out]
        self.bias = Parameter(torch.zeros(out_features)) if _has(bias_attr) else None

    def forward(self, x):
        return F_linear(x, self.weight, self.bias)

    # ppdiffusers/patches/paddle_patch.py:218-228 adds these two properties to paddle.nn.Linear
    in_features = property(lambda self: self.weight.shape[0])
    out_features = property(lambda self: self.weight.shape[1])


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2D(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, padding_mode="zeros",
                 weight_attr=None, bias_attr=None, data_format="NCHW"):
        super().__init__()
        assert data_format == "NCHW" and padding_mode == "zeros"
        kh, kw = _pair(kernel_size)
        self._stride, self._padding, self._dilation, self._groups = stride, padding, dilation, groups
        self._in_channels, self._out_channels, self._kernel_size = in_channels, out_channels, (kh, kw)
        self.weight = Parameter(torch.zeros(out_channels, in_channels // groups, kh, kw))
        self.bias = Parameter(torch.zeros(out_channels)) if _has(bias_attr) else None

    def forward(self, x):
        return F_conv2d(x, self.weight, self.bias, self._stride, self._padding, self._dilation, self._groups)


class GroupNorm(Layer):
    def __init__(self, num_groups, num_channels, epsilon=1e-05, weight_attr=None, bias_attr=None, data_format="NCHW", name=None):
        super().__init__()
        assert data_format == "NCHW"
        self._num_groups, self._epsilon = num_groups, epsilon
        self.weight = Parameter(torch.ones(num_channels)) if _has(weight_attr) else None
        self.bias = Parameter(torch.zeros(num_channels)) if _has(bias_attr) else None

    def forward(self, x):
        return Tensor(TF.group_norm(_u(x), self._num_groups, _u(self.weight), _u(self.bias), self._epsilon))


class LayerNorm(Layer):
    def __init__(self, normalized_shape, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        ns = [normalized_shape] if isinstance(normalized_shape, int) else list(normalized_shape)
        self._normalized_shape, self._epsilon = ns, epsilon
        self.weight = Parameter(torch.ones(ns)) if _has(weight_attr) else None
        self.bias = Parameter(torch.zeros(ns)) if _has(bias_attr) else None

    def forward(self, x):
        return Tensor(TF.layer_norm(_u(x), self._normalized_shape, _u(self.weight), _u(self.bias), self._epsilon))


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None, name=None):
        super().__init__()
        self.weight = Parameter(torch.zeros(num_embeddings, embedding_dim))

    def forward(self, ids):
        return Tensor(TF.embedding(_u(ids).long(), self.weight.t))


class Dropout(Layer):
    def __init__(self, p=0.5, axis=None, mode="upscale_in_train", name=None):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert not self.training or self.p == 0.0, "paddle_shim: dropout only in eval mode / p = 0"
        return x


class Identity(Layer):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x, *a, **k):
        return x


def _act(fn):
    class A(Layer):
        def __init__(self, *a, **k):
            super().__init__()
            self._a, self._k = a, k

        def forward(self, x):
            return fn(x, *self._a, **self._k)
    return A


class AvgPool2D(Layer):
    def __init__(self, kernel_size, stride=None, padding=0, **k):
        super().__init__()
        self._k, self._s, self._p = kernel_size, stride, padding

    def forward(self, x):
        return Tensor(TF.avg_pool2d(_u(x), self._k, self._s, self._p))


def _unbuilt(name):
    class U(Layer):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"paddle_shim: nn.{name} is outside the hot path")
    U.__name__ = name
    return U


# ------------------------------------------------------------------------------------------------------------------ functional
def F_linear(x, weight, bias=None, name=None):
    y = torch.matmul(_u(x), _u(weight))
    return Tensor(y + _u(bias) if bias is not None else y)


def F_conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, data_format="NCHW", name=None):
    assert data_format == "NCHW"
    if isinstance(padding, str):
        padding = padding.lower()
    return Tensor(TF.conv2d(_u(x), _u(weight), _u(bias), stride, padding, dilation, groups))


def F_silu(x, name=None):
    return Tensor(TF.silu(_u(x)))


def F_gelu(x, approximate=False, name=None):
    return Tensor(TF.gelu(_u(x), approximate="tanh" if approximate else "none"))


def F_softmax(x, axis=-1, dtype=None, name=None):
    return Tensor(torch.softmax(_u(x), dim=axis, dtype=_dtype(dtype)))


def F_log_softmax(x, axis=-1, dtype=None, name=None):
    return Tensor(torch.log_softmax(_u(x), dim=axis, dtype=_dtype(dtype)))


def F_dropout(x, p=0.5, axis=None, training=True, mode="upscale_in_train", name=None):
    assert not training or p == 0.0
    return x


def F_interpolate(x, size=None, scale_factor=None, mode="nearest", align_corners=False, align_mode=0, data_format="NCHW", name=None):
    assert data_format == "NCHW"
    if size is not None:
        size = [int(_u(s)) for s in size]
    return Tensor(TF.interpolate(_u(x), size=size, scale_factor=scale_factor, mode=mode, align_corners=None if mode == "nearest" else align_corners))


def F_pad(x, pad, mode="constant", value=0.0, data_format="NCHW", name=None):
    pad = [int(_u(p)) for p in pad]
    t = _u(x)
    if len(pad) == 2 * t.dim():
        # paddle: a full-rank pad list runs from the FIRST axis to the last; torch's runs from the last
        pairs = [pad[2 * i:2 * i + 2] for i in range(t.dim())]
        pad = [v for pr in reversed(pairs) for v in pr]
    return Tensor(TF.pad(t, pad, mode=mode, value=value))


def F_avg_pool2d(x, kernel_size, stride=None, padding=0, **k):
    return Tensor(TF.avg_pool2d(_u(x), kernel_size, stride, padding))


def F_layer_norm(x, normalized_shape, weight=None, bias=None, epsilon=1e-05, name=None):
    ns = [normalized_shape] if isinstance(normalized_shape, int) else list(normalized_shape)
    return Tensor(TF.layer_norm(_u(x), ns, _u(weight), _u(bias), epsilon))


def F_normalize(x, p=2, axis=1, epsilon=1e-12):
    return Tensor(TF.normalize(_u(x), p=p, dim=axis, eps=epsilon))


def F_mish(x):
    return Tensor(TF.mish(_u(x)))


def F_relu(x):
    return Tensor(TF.relu(_u(x)))


def F_scaled_dot_product_attention_(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, training=True,
                                    attention_op=None, name=None):
    """The reference's own "math" branch of its attention dispatcher (ppdiffusers/patches/paddle_patch.py:445-461), restated:
    inputs and output are [batch, tokens, heads, head_dim]; the fused branches (cutlass / flash) compute the same function."""
    assert not is_causal and (dropout_p == 0.0 or not training)
    q, k, v = (_u(t).permute(0, 2, 1, 3) for t in (query, key, value))
    if scale is None:
        scale = 1 / math.sqrt(q.shape[-1])
    s = torch.matmul(q * scale, k.transpose(-1, -2))
    if attn_mask is not None:
        s = s + _u(attn_mask).to(s.dtype)
    return Tensor(torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3))


F_scaled_dot_product_attention = F_scaled_dot_product_attention_   # paddle's own entry point: same layout, scale 1/sqrt(d)


def recompute(fn, *a, **k):
    k.pop("use_reentrant", None)
    return fn(*a, **k)


# ------------------------------------------------------------------------------------------------------------------ module tree
def build_modules():
    """the `paddle` package as module objects (not installed into sys.modules here: reference_runner.install() does that)"""
    me = sys.modules[__name__]
    paddle = types.ModuleType("paddle")
    for k, v in vars(me).items():
        if not k.startswith("_") and k not in ("build_modules", "F_linear") and not k.startswith("F_"):
            setattr(paddle, k, v)
    paddle.bool = bool_
    paddle.__version__ = "0.0.0-torch-shim"
    paddle.dtype = torch.dtype
    nn = types.ModuleType("paddle.nn")
    for n in ("Layer", "LayerList", "Sequential", "Linear", "Conv2D", "GroupNorm", "LayerNorm", "Embedding", "Dropout", "Identity", "AvgPool2D"):
        setattr(nn, n, getattr(me, n))
    nn.Silu, nn.GELU, nn.Mish, nn.ReLU, nn.Sigmoid, nn.Tanh = _act(F_silu), _act(F_gelu), _act(F_mish), _act(F_relu), _act(sigmoid), _act(tanh)
    nn.Silu.__name__, nn.GELU.__name__ = "Silu", "GELU"
    for n in ("Conv1D", "Conv3D", "Conv2DTranspose", "Conv1DTranspose", "Conv3DTranspose", "AvgPool1D", "Upsample", "MultiHeadAttention", "Pad2D", "BCEWithLogitsLoss", "CrossEntropyLoss",
              "MSELoss"):
        setattr(nn, n, _unbuilt(n))
    nn.Parameter = lambda t, trainable=True: Parameter(_u(t), trainable)
    nn.ParameterList = LayerList
    F = types.ModuleType("paddle.nn.functional")
    for k, v in vars(me).items():
        if k.startswith("F_"):
            setattr(F, k[2:], v)
    F.sigmoid, F.tanh = sigmoid, tanh
    init = types.ModuleType("paddle.nn.initializer")
    for n in ("Constant", "Normal", "TruncatedNormal", "XavierUniform", "XavierNormal", "KaimingUniform", "KaimingNormal", "Uniform", "Assign"):
        setattr(init, n, type(n, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None}))
    nn.functional, nn.initializer = F, init
    nn.init = types.ModuleType("paddle.nn.init")   # ppdiffusers' torch-style initialisers: placeholders (parameters are set by name)
    for n in ("normal_", "zeros_", "ones_", "constant_", "xavier_uniform_", "kaiming_uniform_", "trunc_normal_", "uniform_"):
        setattr(nn.init, n, lambda t, *a, **k: t)
    paddle.nn = nn
    dist = types.ModuleType("paddle.distributed")
    fleet = types.ModuleType("paddle.distributed.fleet")
    futils = types.ModuleType("paddle.distributed.fleet.utils")
    futils.recompute = recompute
    fleet.utils, dist.fleet = futils, fleet
    dist.get_world_size = lambda *a, **k: 1
    dist.get_rank = lambda *a, **k: 0
    paddle.distributed = dist
    incubate = types.ModuleType("paddle.incubate")
    paddle.incubate = incubate
    paddle.framework = types.ModuleType("paddle.framework")
    paddle.framework.in_dynamic_mode = in_dynamic_mode
    paddle.device = types.ModuleType("paddle.device")
    paddle.device.is_compiled_with_cuda = lambda: False
    paddle.is_compiled_with_cuda = lambda: False
    paddle.__path__ = []
    amp = types.ModuleType("paddle.amp")
    amp.auto_cast = types.ModuleType("paddle.amp.auto_cast")
    amp.auto_cast.amp_state = lambda: None        # no autocast region is ever active here
    amp.is_float16_supported = amp.is_bfloat16_supported = lambda *a, **k: False
    paddle.amp = amp
    return {"paddle.amp": amp, "paddle.amp.auto_cast": amp.auto_cast,"paddle": paddle, "paddle.nn": nn, "paddle.nn.functional": F, "paddle.nn.initializer": init, "paddle.distributed": dist,
            "paddle.distributed.fleet": fleet, "paddle.distributed.fleet.utils": futils, "paddle.incubate": incubate,
            "paddle.framework": paddle.framework, "paddle.device": paddle.device}

