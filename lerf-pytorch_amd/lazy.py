"""Device-backed stand-in for the numpy arrays that travel between the reference's call sites.

`eltr._worker` (resample/eval_lut_sr.py:541-665, resample/eval_lut_warp.py:100-222) calls FourSimplexInterpFaster 24 times
per image and, between the calls, applies a handful of numpy operations to the results:

    pred = 0; pred += FourSimplexInterpFaster(...)                 (:547-564, 582-619)
    img_lr = np.round(np.clip(pred / avg_factor + bias, 0, norm)).astype(np.float32).transpose((1, 2, 0))   (:573-577)
    np.rot90(img_lr, r); np.pad(..., ((0, pad), (0, pad), (0, 0)), mode="edge").transpose((2, 0, 1))        (:549-553)
    img_hyper[idx, :, :]; resizer.resize(img_lr, ...); np.clip(np.round(img_out).transpose((1, 2, 0)), 0, norm).astype(np.uint8)

With numpy results every call is a device round trip (stage 2 at 1080p: 12 x 149 MB of float64 back to the host, then the
same pixels up again).  `DeviceArray` keeps the values in HBM and answers exactly those operations there, in the dtype
numpy would use (float64 sums of multiples of 1/16, IEEE division, round-half-even: bit-equal), so the unchanged call
sites run stage 1 -> stage 2 -> stage 3 without leaving the device; anything else numpy asks of it (`np.asarray`,
`__array_interface__` for PIL, an unsupported function) materialises the host copy once and proceeds on that -- slower,
never different.  `set_enabled(False)` restores plain numpy results.
"""
from __future__ import annotations

import numpy as np

_ENABLED = True


def set_enabled(flag: bool):
    """True (default): FourSimplexInterpFaster and the numpy resampler classes return DeviceArray; False: numpy arrays."""
    global _ENABLED
    _ENABLED = bool(flag)


def enabled() -> bool:
    return _ENABLED


def _torch():
    import torch
    return torch


_NP2T = None


def _np_to_torch_dtype(dt):
    global _NP2T
    torch = _torch()
    if _NP2T is None:
        _NP2T = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8,
                 np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
                 np.dtype(np.int64): torch.int64, np.dtype(np.bool_): torch.bool, np.dtype(np.float16): torch.float16}
    return _NP2T.get(np.dtype(dt))


def _torch_to_np_dtype(dt):
    torch = _torch()
    return {torch.float64: np.float64, torch.float32: np.float32, torch.uint8: np.uint8, torch.int8: np.int8,
            torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64, torch.bool: np.bool_,
            torch.float16: np.float16}[dt]


HANDLED = {}
_UFUNC_OPS = {np.add: ("__add__", "__radd__"), np.subtract: ("__sub__", "__rsub__"), np.multiply: ("__mul__", "__rmul__"),
              np.true_divide: ("__truediv__", "__rtruediv__")}


def _implements(*funcs):
    def deco(f):
        for fn in funcs:
            HANDLED[fn] = f
        return f
    return deco


def _unwrap(x):
    return x.t if isinstance(x, DeviceArray) else x


def _host(x):
    """DeviceArray -> its numpy copy, recursively through the containers numpy functions take"""
    if isinstance(x, DeviceArray):
        return x.numpy()
    if isinstance(x, (list, tuple)):
        return type(x)(_host(v) for v in x)
    if isinstance(x, dict):
        return {k: _host(v) for k, v in x.items()}
    return x


class DeviceArray(object):
    """An ndarray-shaped view of a torch CUDA tensor.  See the module docstring for what stays on the device."""
    __array_priority__ = 1000.0

    def __init__(self, t):
        self.t = t
        self._np = None

    # ---- ndarray surface
    shape = property(lambda self: tuple(self.t.shape))
    ndim = property(lambda self: self.t.dim())
    size = property(lambda self: self.t.numel())
    dtype = property(lambda self: np.dtype(_torch_to_np_dtype(self.t.dtype)))
    T = property(lambda self: DeviceArray(self.t.permute(*reversed(range(self.t.dim())))))

    def __len__(self):
        return self.t.shape[0]

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%s)" % (self.shape, self.dtype, self.t.device)

    def numpy(self):
        """the host copy (made once; the object is treated as immutable by everything in this package)"""
        if self._np is None:
            self._np = download(self.t)
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    @property
    def __array_interface__(self):
        # PIL.Image.fromarray reads shape / typestr here and then wants the BUFFER protocol of the object when `strides` is
        # None -- which a Python class cannot offer before 3.12; with explicit strides it asks for .tobytes() instead
        a = self.numpy()
        d = dict(a.__array_interface__)
        d["strides"] = a.strides
        return d

    def tobytes(self, *a, **k):
        return self.numpy().tobytes(*a, **k)

    def astype(self, dtype, *a, **k):
        td = _np_to_torch_dtype(dtype)
        if td is None:
            return self.numpy().astype(dtype, *a, **k)
        return DeviceArray(self.t.to(td))             # float -> integer truncates, as numpy's conversion does

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes or axes == (None,):
            axes = tuple(reversed(range(self.t.dim())))
        return DeviceArray(self.t.permute(*axes))

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return DeviceArray(self.t.reshape(*shape))

    def copy(self):
        return DeviceArray(self.t.clone())

    def round(self, decimals=0, out=None):
        if decimals != 0 or out is not None:
            return self.numpy().round(decimals, out)
        return DeviceArray(self.t.round() if self.t.is_floating_point() else self.t)      # half to even, like np.round

    def clip(self, min=None, max=None, out=None, **kw):
        if out is not None or kw:
            return self.numpy().clip(min, max, out, **kw)
        return DeviceArray(self.t.clamp(min=_scalar(min), max=_scalar(max)))

    def __getattr__(self, name):
        # anything else an ndarray offers (max, mean, tobytes, flags ...): on the host copy
        if name.startswith("__") or name in ("t", "_np"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def __getitem__(self, idx):
        try:
            return DeviceArray(self.t[idx])
        except (TypeError, IndexError, RuntimeError):
            return self.numpy()[idx]

    # ---- arithmetic with scalars and other arrays (numpy's result types: float64 stays float64, float32 with python
    #      scalars stays float32)
    def _bin(self, other, op, reflected=False):
        torch = _torch()
        o = other
        if isinstance(o, DeviceArray):
            o = o.t
        elif isinstance(o, np.ndarray):
            if o.ndim == 0:
                o = o.item()
            else:
                td = _np_to_torch_dtype(o.dtype)
                if td is None:
                    return NotImplemented
                o = torch.from_numpy(np.ascontiguousarray(o)).to(self.t.device)
        elif isinstance(o, (np.generic,)):
            o = o.item()
        elif not isinstance(o, (int, float, bool)):
            return NotImplemented
        a, b = (o, self.t) if reflected else (self.t, o)
        return DeviceArray(op(a, b))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
    def __iadd__(self, o): return self._bin(o, lambda a, b: a + b)          # a fresh array: `pred += x` rebinds `pred`
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)

    # Division by a host scalar: torch multiplies by the reciprocal when the divisor is a CPU scalar (one rounding more than
    # numpy's IEEE division: (N / 16) / 3 at N = 48 k + 24 must be exactly k + 0.5 for the round-half-even that follows).
    # The scalar therefore travels as a 0-dim DEVICE tensor of the array's dtype: a true division.
    def _div(self, o, reflected):
        torch = _torch()
        t = self.t if self.t.is_floating_point() else self.t.to(torch.float64)    # numpy: int / x -> float64
        if isinstance(o, (np.generic,)) or (isinstance(o, np.ndarray) and o.ndim == 0):
            o = o.item()
        if isinstance(o, (int, float, bool)):
            o = DeviceArray(torch.full((), float(o), dtype=t.dtype, device=t.device))
        return DeviceArray(t)._bin(o, lambda a, b: torch.true_divide(a, b), reflected)

    def __truediv__(self, o):
        return self._div(o, False)

    def __rtruediv__(self, o):
        return self._div(o, True)

    def __neg__(self): return DeviceArray(-self.t)
    def __abs__(self): return DeviceArray(self.t.abs())

    # comparisons on the device (boolean arrays, like numpy's); a comparison torch cannot form falls back to the host copy
    def _cmp(self, o, op, hop):
        r = self._bin(o, op)
        return hop(self.numpy(), _host(o)) if r is NotImplemented else r

    def __eq__(self, o): return self._cmp(o, lambda a, b: a == b, lambda a, b: a == b)
    def __ne__(self, o): return self._cmp(o, lambda a, b: a != b, lambda a, b: a != b)
    def __lt__(self, o): return self._cmp(o, lambda a, b: a < b, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, lambda a, b: a <= b, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, lambda a, b: a > b, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, lambda a, b: a >= b, lambda a, b: a >= b)
    __hash__ = None

    def __bool__(self):
        return bool(self.numpy())

    def __float__(self):
        return float(self.numpy())

    def __int__(self):
        return int(self.numpy())

    # the rest of the operator table: numpy's own result on the host copy
    def __pow__(self, o): return self.numpy() ** _host(o)
    def __rpow__(self, o): return _host(o) ** self.numpy()
    def __mod__(self, o): return self.numpy() % _host(o)
    def __floordiv__(self, o): return self.numpy() // _host(o)
    def __and__(self, o): return self.numpy() & _host(o)
    def __or__(self, o): return self.numpy() | _host(o)
    def __xor__(self, o): return self.numpy() ^ _host(o)
    def __invert__(self): return ~self.numpy()
    def __matmul__(self, o): return self.numpy() @ _host(o)
    def __iter__(self): return iter(self.numpy())

    # ---- numpy protocol: the handful of functions of the call sites run on the device, everything else on the host copy
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method == "__call__" and not kwargs:
            names = _UFUNC_OPS.get(ufunc)
            if names is not None and len(inputs) == 2:
                a, b = inputs
                r = getattr(a, names[0])(b) if isinstance(a, DeviceArray) else getattr(b, names[1])(a)
                if r is not NotImplemented:
                    return r
            if ufunc is np.rint and len(inputs) == 1:
                return self.round()
        return getattr(ufunc, method)(*_host(inputs), **_host(kwargs))

    def __array_function__(self, func, types, args, kwargs):
        h = HANDLED.get(func)
        if h is not None:
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                return r
        return func(*_host(args), **_host(kwargs))


def _scalar(v):
    if v is None:
        return None
    if isinstance(v, (np.generic, np.ndarray)):
        return v.item()
    return v


@_implements(np.round, np.around)
def _round(a, decimals=0, out=None):
    return a.round(decimals, out) if isinstance(a, DeviceArray) else NotImplemented


@_implements(np.clip)
def _clip(a, a_min=None, a_max=None, out=None, **kw):
    if not isinstance(a, DeviceArray) or out is not None or kw or isinstance(a_min, DeviceArray) or isinstance(a_max, DeviceArray):
        return NotImplemented
    return a.clip(a_min, a_max)


@_implements(np.transpose)
def _transpose(a, axes=None):
    return a.transpose(axes) if axes is not None else a.transpose()


@_implements(np.rot90)
def _rot90(m, k=1, axes=(0, 1)):
    torch = _torch()
    return DeviceArray(torch.rot90(m.t, int(k), [int(axes[0]), int(axes[1])]))


@_implements(np.pad)
def _pad(array, pad_width, mode="constant", **kw):
    """edge / constant(0) padding by index selection (the call sites pad bottom / right with mode="edge", :551-553)"""
    if not isinstance(array, DeviceArray) or kw or mode not in ("edge", "constant"):
        return NotImplemented
    torch = _torch()
    t = array.t
    pw = np.asarray(pad_width)
    if pw.ndim == 0:
        pw = np.tile(pw, (t.dim(), 2))
    elif pw.ndim == 1:
        pw = np.tile(pw.reshape(1, -1), (t.dim(), 1)) if pw.size == 2 else None
    if pw is None or pw.shape != (t.dim(), 2) or (pw < 0).any():
        return NotImplemented
    if mode == "edge":
        for d in range(t.dim()):
            lo, hi = int(pw[d, 0]), int(pw[d, 1])
            if lo or hi:
                idx = torch.arange(-lo, t.shape[d] + hi, device=t.device).clamp_(0, t.shape[d] - 1)
                t = t.index_select(d, idx)
        return DeviceArray(t)
    out = torch.zeros([t.shape[d] + int(pw[d, 0]) + int(pw[d, 1]) for d in range(t.dim())], dtype=t.dtype, device=t.device)
    out[tuple(slice(int(pw[d, 0]), int(pw[d, 0]) + t.shape[d]) for d in range(t.dim()))] = t
    return DeviceArray(out)


@_implements(np.shape)
def _shape(a):
    return a.shape


@_implements(np.ndim)
def _ndim(a):
    return a.ndim


@_implements(np.concatenate)
def _concatenate(arrays, axis=0, out=None, **kw):
    if out is not None or kw or not all(isinstance(a, DeviceArray) for a in arrays):
        return NotImplemented
    torch = _torch()
    return DeviceArray(torch.cat([a.t for a in arrays], dim=int(axis)))


@_implements(np.expand_dims)
def _expand_dims(a, axis):
    return DeviceArray(a.t.unsqueeze(int(axis))) if isinstance(axis, (int, np.integer)) else NotImplemented


def asdevice(a, dtype=None):
    """numpy array (or DeviceArray / CUDA tensor) -> DeviceArray.  One line at the top of a worker (`img_lr = asdevice(img_lr)`)
    moves the rot90 / pad of the FIRST stage to the device as well: the 12 stage-1 calls then upload nothing."""
    torch = _torch()
    if isinstance(a, DeviceArray):
        return a if dtype is None else a.astype(dtype)
    if isinstance(a, torch.Tensor):
        return DeviceArray(a if a.is_cuda else a.cuda())
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
    return DeviceArray(upload(a))


# ---- host -> device
_STAGE = {}          # (nbytes) -> [slots, next]; a slot = (pinned tensor, its numpy view, event of its last copy)
_STAGE_LOCK = __import__("threading").Lock()      # the ring is shared by every caller of the process
_STAGE_MIN = 1 << 20
_STAGE_MAX_BYTES = 256 << 20


def upload(a):
    """contiguous numpy array -> CUDA tensor.  Arrays of 1 MB and more go through a ring of two pinned buffers per size: a
    single-threaded `np.copyto` (0.9 ms per 25 MB) and an asynchronous copy, so the host neither waits for the device work queued
    before nor depends on the runtime's pageable path, which takes 0.55 ms per 25 MB on one host and 2.4 ms on the next
    (profiles/r04_experiments.txt); smaller ones take the plain pageable copy.  (torch's `pinned.copy_(pageable)` is not used: it
    runs on the CPU thread pool, slow and erratic in a container with fewer CPUs than the machine shows.)"""
    torch = _torch()
    a = np.ascontiguousarray(a)
    if a.nbytes < _STAGE_MIN or a.dtype.hasobject:
        return torch.from_numpy(a).cuda()
    tdt = _torch_dtype(torch, a.dtype)                 # raises for dtypes torch does not have, before anything is staged
    with _STAGE_LOCK:
        ring = _STAGE.get(a.nbytes)
        if ring is None:
            if sum(k * 2 for k in _STAGE) + 2 * a.nbytes > _STAGE_MAX_BYTES:
                _STAGE.clear()
            ring = _STAGE[a.nbytes] = [[], 0]
            for _ in range(2):
                t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
                ring[0].append([t, t.numpy(), None])
        slot = ring[0][ring[1] & 1]
        ring[1] += 1
        if slot[2] is not None:
            slot[2].synchronize()
        np.copyto(slot[1], a.reshape(-1).view(np.uint8))
        d = slot[0].cuda(non_blocking=True).view(tdt).reshape(a.shape)
        ev = torch.cuda.Event()
        ev.record()
        slot[2] = ev
    return d


def _torch_dtype(torch, dt):
    return torch.from_numpy(np.empty(0, dtype=dt)).dtype


def download(t):
    """CUDA tensor -> numpy array.  1 MB and more land in a pinned host tensor from torch's caching host allocator (the DMA runs at
    the link's rate: 0.6 ms instead of 2.9 ms for a 25 MB frame) and the array returned is a view of it -- ordinary writable
    memory to its user, handed back to the allocator's cache when the array is dropped."""
    torch = _torch()
    if not t.is_cuda:
        return t.contiguous().numpy()
    t = t.contiguous()
    if t.numel() * t.element_size() < _STAGE_MIN:
        return t.cpu().numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()
