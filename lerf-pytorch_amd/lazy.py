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

import sys
import weakref

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
              np.true_divide: ("__truediv__", "__rtruediv__"),
              np.equal: ("__eq__", "__eq__"), np.not_equal: ("__ne__", "__ne__"), np.less: ("__lt__", "__gt__"),
              np.less_equal: ("__le__", "__ge__"), np.greater: ("__gt__", "__lt__"), np.greater_equal: ("__ge__", "__le__")}


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


_PY_SCALARS = (bool, int, float)


def _result_dtype(ufunc, a, b):
    """numpy's OWN result dtype for `ufunc(a, b)`, by running it on empty arrays of the operands' dtypes (python scalars stay
    python scalars: numpy 2's weak promotion; numpy scalars stay strongly typed).  None when numpy would raise (e.g.
    uint8 + 300) -- the caller then takes the host path and gets numpy's own error."""
    def probe(x):
        if isinstance(x, DeviceArray):
            return np.empty(0, dtype=x.dtype)
        if isinstance(x, np.ndarray):
            return np.empty(0, dtype=x.dtype) if x.ndim else x[()]
        return x
    try:
        return ufunc(probe(a), probe(b)).dtype
    except Exception:
        return None


class DeviceArray(object):
    """An ndarray-shaped view of a torch tensor (a CUDA tensor in product use).  See the module docstring for what stays on
    the device.  Contract: every operation gives numpy's value AND dtype; whatever the device path cannot guarantee that for
    runs on the host copy (numpy's own result).  Views made by transpose / reshape / basic indexing share their base's storage
    as in numpy: a write through any of them (`a[idx] = v`, `a += v`, `out=a`) is seen by all, and drops every cached host copy.
    Not an `np.ndarray` subclass (`isinstance(x, np.ndarray)` is False -- INTEGRATION.md section 2)."""
    __array_priority__ = 1000.0

    def __init__(self, t, base=None, u8=None):
        self.t = t
        self._np = None
        self._np_ver = -1
        self._vc = base._vc if base is not None else [0]      # version cell shared by all views of one storage
        # provenance: (uint8 tensor, scale, storage version) with self == uint8 / scale exactly (the stage outputs: feat = an
        # integer 0..255, hyper = numerator / 255); lets a resampler fed with them take the uint8 production kernels.  Valid only
        # while the storage has not been written since (exact_u8()).
        self._u8 = (u8[0], float(u8[1]), self._vc[0]) if u8 is not None else None

    def exact_u8(self):
        """(uint8 tensor of this array's shape, scale) with array == uint8 / scale exactly, or None"""
        u = getattr(self, "_u8", None)
        return (u[0], u[1]) if u is not None and u[2] == self._vc[0] else None

    # ---- ndarray surface
    shape = property(lambda self: tuple(self.t.shape))
    ndim = property(lambda self: self.t.dim())
    size = property(lambda self: self.t.numel())
    dtype = property(lambda self: np.dtype(_torch_to_np_dtype(self.t.dtype)))
    itemsize = property(lambda self: self.t.element_size())
    nbytes = property(lambda self: self.t.numel() * self.t.element_size())
    T = property(lambda self: self.transpose())

    def __len__(self):
        if self.t.dim() == 0:
            raise TypeError("len() of unsized object")          # numpy's words; torch's sequence walk relies on the TypeError
        return self.t.shape[0]

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%s)" % (self.shape, self.dtype, self.t.device)

    def _touch(self):
        """the storage was written: every cached host copy of it is stale"""
        self._vc[0] += 1
        self._np = None

    def numpy(self):
        """the host copy: made once per version of the storage, READ-ONLY (a write into it could not reach the device; numpy
        raises instead of losing it -- `np.array(x)` / `x.copy()` give writable arrays)"""
        if self._np is None or self._np_ver != self._vc[0]:
            a = download(self.t)
            a.flags.writeable = False
            self._np, self._np_ver = a, self._vc[0]
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy(order="K") if copy else a

    @property
    def __array_interface__(self):
        # PIL.Image.fromarray reads shape / typestr here and then wants the BUFFER protocol of the object when `strides` is
        # None -- which a Python class cannot offer before 3.12; with explicit strides it asks for .tobytes() instead
        a = self.numpy()
        d = dict(a.__array_interface__)
        d["strides"] = a.strides
        return d

    # torch.Tensor(x) / torch.tensor(x) / torch.as_tensor(x) (the warp harness builds its mPSNR operands that way,
    # resample/eval_lut_warp.py:225-233) look for DLPack before they walk an object as a sequence.  The export is a HOST tensor
    # -- to torch this object is what it is to numpy, a CPU array -- and a private copy, so that nothing aliases the cache.
    def __dlpack__(self, *a, **k):
        return _torch().from_numpy(self.numpy().copy(order="K")).__dlpack__()

    def __dlpack_device__(self):
        return (1, 0)                                           # kDLCPU

    def tobytes(self, *a, **k):
        return self.numpy().tobytes(*a, **k)

    def tolist(self):
        return self.numpy().tolist()

    def item(self, *a):
        return self.numpy().item(*a)

    def astype(self, dtype, *a, **k):
        td = _np_to_torch_dtype(dtype)
        if td is None or a or k:
            return self.numpy().astype(dtype, *a, **k)
        t = self.t
        if t.is_floating_point() and not td.is_floating_point and td != _torch().bool:
            # float -> integer: numpy truncates toward zero and wraps modulo 2^n through int64 for out-of-range values on
            # this platform; torch's direct conversion saturates or is undefined there.  Via int64 both agree for every
            # finite value below 2^63 (the call sites convert clip(round(.), 0, 255)).
            t = t.to(_torch().int64)
        return DeviceArray(t.to(td, copy=True))

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes or axes == (None,):
            axes = tuple(reversed(range(self.t.dim())))
        u = self.exact_u8()
        return DeviceArray(self.t.permute(*axes), base=self, u8=(u[0].permute(*axes), u[1]) if u else None)

    def reshape(self, *shape, **kw):
        if kw:
            return self.numpy().reshape(*shape, **kw)
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return DeviceArray(self.t.reshape(*shape), base=self)     # a view where torch can make one, like numpy

    def copy(self, *a, **k):
        return DeviceArray(self.t.clone())

    def round(self, decimals=0, out=None):
        if decimals != 0 or out is not None:
            return _with_out(lambda: self.numpy().round(decimals), out)
        return DeviceArray(self.t.round() if self.t.is_floating_point() else self.t.clone())      # half to even, like np.round

    def clip(self, min=None, max=None, out=None, **kw):
        lo, hi = _scalar(min), _scalar(max)
        ok = (not kw and (lo is not None or hi is not None) and all(v is None or isinstance(v, _PY_SCALARS) for v in (lo, hi))
              and _result_dtype(np.add, self, 0 if lo is None else lo) == self.dtype
              and _result_dtype(np.add, self, 0 if hi is None else hi) == self.dtype)
        if not ok or out is not None:
            return _with_out(lambda: self.numpy().clip(_host(min), _host(max), **kw), out)
        return DeviceArray(self.t.clamp(min=lo, max=hi))

    def __getattr__(self, name):
        # anything else an ndarray offers (max, mean, flags ...): on the (read-only) host copy
        if name.startswith("__") or name in ("t", "_np", "_np_ver", "_vc"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    # ---- indexing.  On the device: basic indices (a view, as in numpy) and ONE advanced index among slices (the call
    #      sites' img_hyper[[0, 3, 6], :, :] and a[mask]); combinations whose axis placement rules are subtle: host copy.
    def _index(self, idx):
        """-> (torch index, is_view) or None"""
        torch = _torch()
        items = idx if isinstance(idx, tuple) else (idx,)
        out, adv, ints = [], 0, 0
        for it in items:
            if isinstance(it, DeviceArray):
                it = it.t
            if it is None or it is Ellipsis or isinstance(it, slice):
                out.append(it)
            elif isinstance(it, (int, np.integer)) and not isinstance(it, (bool, np.bool_)):
                out.append(int(it))
                ints += 1
            elif isinstance(it, (list, np.ndarray, torch.Tensor)):
                if isinstance(it, list):
                    it = np.asarray(it)
                    if it.dtype.kind not in "iub" or it.ndim != 1:
                        return None
                if isinstance(it, np.ndarray):
                    if it.dtype.kind not in "iub":
                        return None
                    it = torch.from_numpy(np.ascontiguousarray(it if it.dtype.kind == "b" else it.astype(np.int64))).to(self.t.device)
                elif it.dtype not in (torch.bool, torch.int64, torch.int32, torch.uint8, torch.int16, torch.int8):
                    return None
                elif it.dtype not in (torch.bool, torch.int64):
                    it = it.to(torch.int64)
                out.append(it.to(self.t.device))
                adv += 1
            else:
                return None
        if adv > 1 or (adv == 1 and ints):
            return None
        return (tuple(out) if isinstance(idx, tuple) else out[0]), adv == 0

    def __getitem__(self, idx):
        ti = self._index(idx)
        if ti is None:
            return self.numpy()[_host(idx)]
        try:
            r = self.t[ti[0]]
        except (TypeError, IndexError, RuntimeError, ValueError):
            return self.numpy()[_host(idx)]                                     # numpy's own error for a bad index
        if r.dim() == 0 and ti[1] and not _has_ellipsis_or_none(idx):
            return self.dtype.type(r.item())                                    # a[i, j, k] is a scalar in numpy
        u = self.exact_u8()
        return DeviceArray(r, base=self if ti[1] else None, u8=(u[0][ti[0]], u[1]) if u else None)

    def __setitem__(self, idx, value):
        torch = _torch()
        _flush_dependents(self._vc)                                             # pending results that READ this storage run first
        ti = self._index(idx)
        v = value
        if isinstance(v, DeviceArray):
            v = v.t
        elif isinstance(v, np.ndarray):
            v = v[()] if v.ndim == 0 else (torch.from_numpy(np.ascontiguousarray(v)).to(self.t.device)
                                          if _np_to_torch_dtype(v.dtype) is not None else None)
        if isinstance(v, np.generic):
            v = v.item()
        dev_ok = ti is not None and (isinstance(v, _PY_SCALARS) or isinstance(v, torch.Tensor))
        if dev_ok and isinstance(v, _PY_SCALARS) and not self.t.is_floating_point():
            # numpy 2 refuses out-of-range python integers for integer arrays and truncates floats: host rules
            dev_ok = isinstance(v, (bool, int)) and np.can_cast(np.min_scalar_type(v), self.dtype, "safe")
        if dev_ok and isinstance(v, torch.Tensor) and v.is_floating_point() and not self.t.is_floating_point():
            v = v.to(torch.int64)                                               # numpy's float -> int conversion (see astype)
        if dev_ok:
            try:
                self.t[ti[0]] = v
                self._touch()
                return
            except (TypeError, IndexError, RuntimeError, ValueError):
                pass
        h = self.numpy().copy()
        h[_host(idx)] = _host(value)                                            # numpy's own semantics and errors
        self.t.copy_(torch.from_numpy(h).to(self.t.device))
        self._touch()

    # ---- arithmetic.  The result dtype is numpy's (`_result_dtype`); the operands are cast to it on the device, which
    #      is what numpy's loops do for + - * and comparisons never reach this path with integer arrays.
    def _operand(self, o, rd):
        """other operand -> torch tensor / python scalar usable with a tensor of numpy dtype `rd`; NotImplemented if none"""
        torch = _torch()
        td = _np_to_torch_dtype(rd)
        if isinstance(o, DeviceArray):
            return o.t.to(td)
        if isinstance(o, np.ndarray):
            if o.ndim == 0:
                o = o[()]
            elif _np_to_torch_dtype(o.dtype) is None:
                return NotImplemented
            else:
                return torch.from_numpy(np.ascontiguousarray(o)).to(self.t.device).to(td)
        if isinstance(o, np.generic):
            o = o.item()
        if isinstance(o, _PY_SCALARS):
            if rd.kind == "f":
                # as a 0-dim device tensor of the result dtype: the value numpy's loop would see (a python float rounded to
                # float32 for a float32 array), and torch treats it as an array operand, not as a reciprocal / opmath scalar
                return torch.full((), float(o), dtype=td, device=self.t.device)
            return o
        return NotImplemented

    def _bin(self, other, ufunc, op, reflected=False, inplace=False):
        if isinstance(other, LazyArray) and other.pending and ufunc is np.add:
            r = other._fold_into_add(self, reflected, inplace)                  # `pred += FourSimplexInterpFaster(...)`: one launch
            if r is not None:
                return r
        if isinstance(self, LazyArray) and self.pending and self._recipe:
            if reflected and ufunc is np.add and isinstance(other, (int, float)) and not isinstance(other, bool) and other == 0 \
                    and self._recipe[0] == "interp" and sys.getrefcount(self) <= _TEMP_REFS:
                # `pred = 0; pred += F(...)`: 0 + x is x for every value the pass produces (no negative zeros) and the temporary
                # has no other owner (CPython reference count, calibrated at import): the pass becomes the sum's first term
                return self._start_sum()
            if not inplace and self._recipe[0] in ("sum", "expr") and isinstance(other, (int, float)) and not isinstance(other, bool) \
                    and ufunc in _EPI_UFUNCS and (not reflected or ufunc in (np.add, np.multiply)):
                r = self._extend(_EPI_UFUNCS[ufunc], float(other))
                if r is not None:
                    return r
        if inplace:
            _flush_dependents(self._vc)
        rd = _result_dtype(ufunc, other, self) if reflected else _result_dtype(ufunc, self, other)
        if rd is None or _np_to_torch_dtype(rd) is None or not isinstance(other, (DeviceArray, np.ndarray, np.generic) + _PY_SCALARS) \
                or rd.kind not in "fiub" or (rd.kind != "f" and ufunc is np.true_divide):
            return self._host_bin(other, ufunc, reflected, inplace)
        if rd.kind != "f" and not (ufunc in (np.add, np.subtract, np.multiply) and isinstance(other, (DeviceArray, np.ndarray))):
            return self._host_bin(other, ufunc, reflected, inplace)             # integer arrays with scalars: numpy's range checks
        o = self._operand(other, rd)
        if o is NotImplemented:
            return self._host_bin(other, ufunc, reflected, inplace)
        if inplace:
            if rd != self.dtype:
                return self._host_bin(other, ufunc, reflected, inplace)         # numpy's casting rule decides (usually an error)
            if hasattr(o, "untyped_storage") and o.numel() > 1 and o.untyped_storage().data_ptr() == self.t.untyped_storage().data_ptr():
                o = o.clone()                                                   # `a -= a[0]`: numpy reads an overlapping operand before it writes
            try:
                op(self.t, o, out=self.t)
            except RuntimeError:                                                # the operand does not broadcast INTO self
                return self._host_bin(other, ufunc, reflected, inplace)
            self._touch()
            return self
        a = self.t.to(_np_to_torch_dtype(rd))
        try:
            res = DeviceArray(op(o, a) if reflected else op(a, o))
        except RuntimeError:
            return self._host_bin(other, ufunc, reflected, inplace)
        if ufunc is np.true_divide and not reflected and isinstance(other, (int, float)) and not isinstance(other, bool) and other == 255 \
                and rd == np.float32:
            u = self.exact_u8()                                                 # hyper = numerators / 255 (:628)
            if u is not None and u[1] == 1.0:
                res._u8 = (u[0], 255.0, res._vc[0])
        return res

    def _host_bin(self, other, ufunc, reflected, inplace):
        if inplace:
            h = self.numpy().copy()
            ufunc(h, _host(other), out=h)                                       # numpy's own casting errors
            self.t.copy_(_torch().from_numpy(h).to(self.t.device))
            self._touch()
            return self
        return ufunc(_host(other), self.numpy()) if reflected else ufunc(self.numpy(), _host(other))

    def _t(name):
        return lambda a, b, out=None: getattr(_torch(), name)(a, b, out=out) if out is not None else getattr(_torch(), name)(a, b)

    def __add__(self, o): return self._bin(o, np.add, DeviceArray._add)
    def __radd__(self, o): return self._bin(o, np.add, DeviceArray._add, True)
    def __iadd__(self, o): return self._bin(o, np.add, DeviceArray._add, inplace=True)     # in place, like numpy
    def __sub__(self, o): return self._bin(o, np.subtract, DeviceArray._sub)
    def __rsub__(self, o): return self._bin(o, np.subtract, DeviceArray._sub, True)
    def __isub__(self, o): return self._bin(o, np.subtract, DeviceArray._sub, inplace=True)
    def __mul__(self, o): return self._bin(o, np.multiply, DeviceArray._mul)
    def __rmul__(self, o): return self._bin(o, np.multiply, DeviceArray._mul, True)
    def __imul__(self, o): return self._bin(o, np.multiply, DeviceArray._mul, inplace=True)
    # Division: IEEE on the device because the divisor is a device TENSOR (`_operand`): torch multiplies by the reciprocal
    # when the divisor is a host scalar -- one rounding more than numpy's division ((N / 16) / 3 at N = 48 k + 24 must be
    # exactly k + 0.5 for the round-half-even that follows).
    def __truediv__(self, o): return self._bin(o, np.true_divide, DeviceArray._div)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, DeviceArray._div, True)
    def __itruediv__(self, o): return self._bin(o, np.true_divide, DeviceArray._div, inplace=True)

    def __neg__(self): return DeviceArray(-self.t) if self.t.is_floating_point() else -self.numpy()
    def __pos__(self): return self.copy()
    def __abs__(self): return DeviceArray(self.t.abs()) if self.t.is_floating_point() else abs(self.numpy())

    # comparisons: boolean arrays like numpy's; on the device for floating-point arrays against arrays / scalars (the warp
    # harness' `mask_output == 255`), else numpy's own result (python integers out of an integer dtype's range etc.)
    def _cmp(self, o, ufunc, op):
        torch = _torch()
        if self.t.is_floating_point() and isinstance(o, (DeviceArray, np.ndarray, np.generic) + _PY_SCALARS):
            ct = _result_dtype(np.add, self, o)                                # the common type numpy compares in
            if ct is not None and ct.kind == "f" and _np_to_torch_dtype(ct) is not None:
                b = self._operand(o, ct)
                if b is not NotImplemented:
                    try:
                        return DeviceArray(op(self.t.to(_np_to_torch_dtype(ct)), b))
                    except RuntimeError:
                        pass
        return ufunc(self.numpy(), _host(o))

    def __eq__(self, o): return self._cmp(o, np.equal, lambda a, b: a == b)
    def __ne__(self, o): return self._cmp(o, np.not_equal, lambda a, b: a != b)
    def __lt__(self, o): return self._cmp(o, np.less, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, np.less_equal, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, np.greater, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, np.greater_equal, lambda a, b: a >= b)
    __hash__ = None

    def __bool__(self):
        return bool(self.numpy())

    def __float__(self):
        return float(self.numpy())

    def __int__(self):
        return int(self.numpy())

    def __index__(self):
        return self.numpy().__index__()

    # the rest of the operator table: numpy's own result on the host copy
    def __pow__(self, o): return self.numpy() ** _host(o)
    def __rpow__(self, o): return _host(o) ** self.numpy()
    def __mod__(self, o): return self.numpy() % _host(o)
    def __rmod__(self, o): return _host(o) % self.numpy()
    def __floordiv__(self, o): return self.numpy() // _host(o)
    def __rfloordiv__(self, o): return _host(o) // self.numpy()
    def __and__(self, o): return self.numpy() & _host(o)
    def __rand__(self, o): return _host(o) & self.numpy()
    def __or__(self, o): return self.numpy() | _host(o)
    def __ror__(self, o): return _host(o) | self.numpy()
    def __xor__(self, o): return self.numpy() ^ _host(o)
    def __rxor__(self, o): return _host(o) ^ self.numpy()
    def __invert__(self): return ~self.numpy()
    def __matmul__(self, o): return self.numpy() @ _host(o)
    def __rmatmul__(self, o): return _host(o) @ self.numpy()
    def __iter__(self):
        if self.t.dim() == 0:
            raise TypeError("iteration over a 0-d array")
        return iter(self.numpy())

    # ---- numpy protocol: the handful of functions of the call sites run on the device, everything else on the host copy
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        out = kwargs.get("out")
        if out is not None and any(isinstance(o, DeviceArray) for o in out):
            # results INTO device arrays: numpy's value (host), then written through to the device storage
            kw = {k: _host(v) for k, v in kwargs.items() if k != "out"}
            r = getattr(ufunc, method)(*_host(inputs), **kw)
            rs = r if isinstance(r, tuple) else (r,)
            for o, v in zip(out, rs):
                if o is not None:
                    o[...] = v
            got = tuple(o if o is not None else v for o, v in zip(out, rs))
            return got if isinstance(r, tuple) else got[0]
        if method == "__call__" and not kwargs:
            names = _UFUNC_OPS.get(ufunc)
            if names is not None and len(inputs) == 2:
                a, b = inputs
                return getattr(a, names[0])(b) if isinstance(a, DeviceArray) else getattr(b, names[1])(a)
            if ufunc is np.rint and len(inputs) == 1:
                return self.round()
        return getattr(ufunc, method)(*_host(inputs), **_host(kwargs))

    def __array_function__(self, func, types, args, kwargs):
        h = HANDLED.get(func)
        if h is not None:
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                return r
        out = kwargs.get("out")
        if isinstance(out, DeviceArray):
            return _with_out(lambda: func(*_host(args), **{k: _host(v) for k, v in kwargs.items() if k != "out"}), out)
        return func(*_host(args), **_host(kwargs))


_PENDING = {}                         # id -> weak reference: LazyArrays that have not run yet (arrays are unhashable, like numpy's)


def _temp_refcount():
    """what sys.getrefcount reports inside _bin for an operand that only the running statement holds (`x = 0; x += f()`),
    measured on this interpreter with the same call depth (__radd__ -> _bin) instead of assumed"""
    seen = []

    class Probe(object):
        def _bin(self, o):
            seen.append(sys.getrefcount(self))
            return self

        def __radd__(self, o):
            return self._bin(o)
    x = 0
    x += Probe()
    return seen[0]


_TEMP_REFS = _temp_refcount()


def _flush_dependents(vc):
    """a storage is about to be written: every pending result that reads it runs first (numpy would have computed it already)"""
    if not _PENDING:
        return
    for ref in list(_PENDING.values()):
        lz = ref()
        if lz is not None and lz.pending and any(d._vc is vc for d in lz._deps):
            lz.t


class LazyArray(DeviceArray):
    """A DeviceArray whose tensor does not exist yet: shape and dtype are known, `make()` produces the tensor when anything
    asks for it (`.t`).  Two kinds are made, both of the call sites' statement sequence (resample/eval_lut_sr.py:549-564):

      np.rot90(img, r) -> np.pad(..., ((0, p), (0, p), (0, 0)), mode="edge") -> .transpose((2, 0, 1))
          recipe ("rot" | "rotpad" | "rotpad_chw", base, k, ph, pw): FourSimplexInterpFaster recognises the chain and runs the
          pass on `base` itself with the pattern's offsets rotated k times and clamped coordinates -- no rotated, no padded copy;
      FourSimplexInterpFaster(...)  recipe ("interp", run): `pred += result` launches the pass ONCE, adding into pred's planes
          (lerf_lut_interp_ex, LERF_INTERP_ACCUMULATE) instead of a launch, a 50 / 150 MB result and an add kernel.

    Used in any other way it materialises and is an ordinary DeviceArray from then on (the old statement-by-statement path, same
    values).  `deps`: the DeviceArrays whose storage `make` reads -- a write to one of them runs every pending reader first."""

    def __init__(self, shape, np_dtype, deps, make, recipe=None):
        self._t = None
        self._shape = tuple(int(v) for v in shape)
        self._dtype = np.dtype(np_dtype)
        self._deps = list(deps)
        self._make = make
        self._recipe = recipe
        self._np = None
        self._np_ver = -1
        self._vc = [0]
        key = id(self)
        _PENDING[key] = weakref.ref(self, lambda _r, k=key: _PENDING.pop(k, None))

    pending = property(lambda self: self._t is None)

    def _get_t(self):
        if self._t is None:
            t = self._make()
            self._t, self._make, self._deps = t, None, []
            _PENDING.pop(id(self), None)
        return self._t

    def _set_t(self, v):
        self._t = v
    t = property(_get_t, _set_t)
    shape = property(lambda self: self._shape if self._t is None else tuple(self._t.shape))
    ndim = property(lambda self: len(self._shape) if self._t is None else self._t.dim())
    size = property(lambda self: int(np.prod(self._shape)) if self._t is None else self._t.numel())
    dtype = property(lambda self: self._dtype if self._t is None else np.dtype(_torch_to_np_dtype(self._t.dtype)))

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if self.pending and self._recipe and self._recipe[0] == "rchain" and tuple(axes) == (1, 2, 0) and len(self._shape) == 3:
            r = self._rchain("hwc", (self._shape[1], self._shape[2], self._shape[0]))
            if r is not None:
                return r
        if self.pending and self._recipe and self._recipe[0] == "rotpad" and tuple(axes) == (2, 0, 1):
            _, base, k, ph, pw = self._recipe
            me = self
            return LazyArray((self._shape[2], self._shape[0], self._shape[1]), self._dtype, self._deps,
                             lambda: me.t.permute(2, 0, 1), ("rotpad_chw", base, k, ph, pw))
        return DeviceArray.transpose(self, *axes)

    # ---- the sum of passes and the stage epilogue (resample/eval_lut_sr.py:547-577): int16 numerators until the caller's
    #      `.astype(np.float32)`, one fused launch for `np.round(np.clip(pred / n + bias, 0, norm)).astype(np.float32)`
    def _start_sum(self):
        """pending pass -> pending sum holding its int16 numerators (value * 2^interval): 2 bytes per element instead of 8"""
        torch = _torch()
        _, run, interval, dev = self._recipe
        acc = torch.empty(self._shape, dtype=torch.int16, device=dev)
        run(acc, False)
        return _sum_array(acc, interval, 127 << interval)

    def _extend(self, op, *operands):
        """pending sum / expression + one more float64 step of the epilogue grammar -> pending expression; None: not ours"""
        kind = self._recipe[0]
        acc, interval = self._recipe[1], self._recipe[2]
        steps = list(self._recipe[3]) if kind == "expr" else []
        if len(steps) >= 8:
            return None
        steps.append((op,) + tuple(float(v) for v in operands))
        return _expr_array(acc, interval, steps)

    # ---- the worker's last statement on a resampler result (resample/eval_lut_sr.py:663-665):
    #      np.clip(np.round(out).transpose((1, 2, 0)), 0, norm).astype(np.uint8)
    def _rchain(self, step, shape=None):
        root, steps = (self, ()) if self._recipe[0] == "resize" else (self._recipe[1], self._recipe[2])
        steps = steps + (step,)
        if steps != _RESIZE_TAIL[:len(steps)]:
            return None
        return LazyArray(shape or self._shape, np.float64, [root], lambda: _apply_tail(root, steps).t, ("rchain", root, steps))

    def clip(self, min=None, max=None, out=None, **kw):
        if self.pending and self._recipe and self._recipe[0] == "rchain" and out is None and not kw and (min, max) == (0, 255) \
                and all(isinstance(v, int) and not isinstance(v, bool) for v in (min, max)):
            r = self._rchain("clip")
            if r is not None:
                return r
        if self.pending and self._recipe and self._recipe[0] in ("sum", "expr") and out is None and not kw \
                and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in (min, max)):
            r = self._extend("clip", min, max)
            if r is not None:
                return r
        return DeviceArray.clip(self, min, max, out=out, **kw)

    def round(self, decimals=0, out=None):
        if self.pending and self._recipe and self._recipe[0] == "resize" and decimals == 0 and out is None:
            r = self._rchain("round")
            if r is not None:
                return r
        if self.pending and self._recipe and self._recipe[0] in ("sum", "expr") and decimals == 0 and out is None:
            r = self._extend("round")
            if r is not None:
                return r
        return DeviceArray.round(self, decimals, out)

    def astype(self, dtype, *a, **k):
        if self.pending and self._recipe and self._recipe[0] == "rchain" and self._recipe[2] == _RESIZE_TAIL and not a and not k \
                and np.dtype(dtype) == np.uint8 and self._recipe[1].pending:
            return DeviceArray(self._recipe[1]._recipe[1]())
        if self.pending and self._recipe and self._recipe[0] in ("sum", "expr") and not a and not k and np.dtype(dtype) == np.float32:
            from . import ops
            kind, acc, interval = self._recipe[0], self._recipe[1], self._recipe[2]
            steps = self._recipe[3] if kind == "expr" else []
            code = {"div": ops.EPI_DIV, "mul": ops.EPI_MUL, "add": ops.EPI_ADD, "clip": ops.EPI_CLIP, "round": ops.EPI_ROUND}
            f = ops.numer_epilogue(acc, interval, [(code[st[0]],) + tuple(st[1:]) for st in steps])
            kinds = [st[0] for st in steps[-2:]]
            clip = [st for st in steps[-2:] if st[0] == "clip"]
            if sorted(kinds) == ["clip", "round"] and clip[0][1] >= 0 and clip[0][2] <= 255:
                return DeviceArray(f, u8=(f.to(_torch().uint8), 1.0))           # integers 0..255: the stage's uint8 output, exactly
            return DeviceArray(f)
        return DeviceArray.astype(self, dtype, *a, **k)

    def _fold_into_add(self, target, reflected, inplace):
        """`target += self` / `target + self` for a pending pass: run it into (a copy of) target's planes; None = not this case"""
        if not (self._recipe and self._recipe[0] == "interp") or not isinstance(target, DeviceArray) or target is self:
            return None
        torch = _torch()
        if isinstance(target, LazyArray) and target.pending and target._recipe and target._recipe[0] == "sum":
            # the sum is still int16 numerators: add this pass's numerators (same interval, no overflow: |n| <= 127 q per pass)
            _, acc, interval, bound = target._recipe
            mine = self._recipe[2]
            if inplace and mine == interval and tuple(acc.shape) == self._shape and bound + (127 << interval) <= 32767 \
                    and self._recipe[1](acc, True):
                target._recipe = ("sum", acc, interval, bound + (127 << interval))
                return target
        tt = target.t
        if tt.dtype != torch.float64 or tuple(tt.shape) != self._shape or not tt.is_cuda or not tt.is_contiguous():
            return None
        run = self._recipe[1]
        if inplace:
            _flush_dependents(target._vc)
            if not run(tt, True):
                return None
            target._touch()
            return target
        out = tt.clone()
        return DeviceArray(out) if run(out, True) else None


_EPI_UFUNCS = {np.true_divide: "div", np.multiply: "mul", np.add: "add"}
_RESIZE_TAIL = ("round", "hwc", "clip")


def _apply_tail(root, steps):
    """the steps of the worker's last statement one at a time on the float64 result (the statement-by-statement path)"""
    x = DeviceArray(root.t)
    for st in steps:
        x = DeviceArray.round(x) if st == "round" else (DeviceArray.transpose(x, 1, 2, 0) if st == "hwc" else DeviceArray.clip(x, 0, 255))
    return x



def _sum_array(acc, interval, bound):
    """pending float64 array = acc / 2^interval (exact), kept as the int16 numerators"""
    torch = _torch()
    return LazyArray(acc.shape, np.float64, [], lambda: acc.to(torch.float64) / float(1 << interval), ("sum", acc, interval, bound))


def _apply_steps(x, steps):
    """the float64 steps one at a time through the ordinary DeviceArray operations (the statement-by-statement path)"""
    for st in steps:
        if st[0] == "div":
            x = x / st[1]
        elif st[0] == "mul":
            x = x * st[1]
        elif st[0] == "add":
            x = x + st[1]
        elif st[0] == "clip":
            x = DeviceArray.clip(x, st[1], st[2])
        else:
            x = DeviceArray.round(x)
    return x


def _expr_array(acc, interval, steps):
    torch = _torch()
    return LazyArray(acc.shape, np.float64, [],
                     lambda: _apply_steps(DeviceArray(acc.to(torch.float64) / float(1 << interval)), steps).t, ("expr", acc, interval, steps))


for _n, _f in (("_add", "add"), ("_sub", "sub"), ("_mul", "mul"), ("_div", "true_divide")):
    setattr(DeviceArray, _n, staticmethod(DeviceArray._t(_f)))
del DeviceArray._t


def _has_ellipsis_or_none(idx):
    items = idx if isinstance(idx, tuple) else (idx,)
    return any(it is None or it is Ellipsis for it in items)


def _with_out(compute, out):
    """numpy's result of `compute()`; into `out` when one is given (a DeviceArray is written through to the device)"""
    r = compute()
    if out is None:
        return r
    out[...] = r
    return out


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
    if not isinstance(a, DeviceArray) or kw or isinstance(a_min, DeviceArray) or isinstance(a_max, DeviceArray):
        return NotImplemented
    return a.clip(a_min, a_max, out=out)


@_implements(np.transpose)
def _transpose(a, axes=None):
    return a.transpose(axes) if axes is not None else a.transpose()


@_implements(np.rot90)
def _rot90(m, k=1, axes=(0, 1)):
    """a fresh array (numpy returns a view: a write through the rotated array does not reach `m` here).  [H, W, C] images rotated
    in their first two axes come back lazy (LazyArray): the call sites only ever pad them and hand them to the LUT pass."""
    if not isinstance(m, DeviceArray):
        return NotImplemented
    torch = _torch()
    k, ax = int(k), (int(axes[0]), int(axes[1]))
    if m.ndim == 3 and ax == (0, 1) and not (isinstance(m, LazyArray) and m.pending) and m.t.is_cuda:
        H, W, Cn = m.shape
        shape = (H, W, Cn) if k % 2 == 0 else (W, H, Cn)
        return LazyArray(shape, m.dtype, [m], lambda: torch.rot90(m.t, k, [0, 1]), ("rot", m, k % 4, 0, 0))
    return DeviceArray(torch.rot90(m.t, k, [ax[0], ax[1]]))


def _pad_now(t, pw, mode):
    torch = _torch()
    if mode == "edge":
        for d in range(t.dim()):
            lo, hi = int(pw[d, 0]), int(pw[d, 1])
            if lo or hi:
                idx = torch.arange(-lo, t.shape[d] + hi, device=t.device).clamp_(0, t.shape[d] - 1)
                t = t.index_select(d, idx)
        return t
    out = torch.zeros([t.shape[d] + int(pw[d, 0]) + int(pw[d, 1]) for d in range(t.dim())], dtype=t.dtype, device=t.device)
    out[tuple(slice(int(pw[d, 0]), int(pw[d, 0]) + t.shape[d]) for d in range(t.dim()))] = t
    return out


@_implements(np.pad)
def _pad(array, pad_width, mode="constant", **kw):
    """edge / constant(0) padding by index selection (the call sites pad bottom / right with mode="edge", :551-553)"""
    if not isinstance(array, DeviceArray) or kw or mode not in ("edge", "constant"):
        return NotImplemented
    pw = np.asarray(pad_width)
    nd = array.ndim
    if pw.ndim == 0:
        pw = np.tile(pw, (nd, 2))
    elif pw.ndim == 1:
        pw = np.tile(pw.reshape(1, -1), (nd, 1)) if pw.size == 2 else None
    if pw is None or pw.shape != (nd, 2) or (pw < 0).any():
        return NotImplemented
    if mode == "edge" and nd == 3 and not pw[:, 0].any() and pw[2, 1] == 0:
        # the call sites' form: [H, W, C] padded at the bottom / right only -- lazy, see LazyArray
        if isinstance(array, LazyArray) and array.pending and array._recipe and array._recipe[0] == "rot":
            base, k = array._recipe[1], array._recipe[2]
        elif not (isinstance(array, LazyArray) and array.pending) and array.t.is_cuda:
            base, k = array, 0
        else:
            base = None
        if base is not None:
            ph, pww = int(pw[0, 1]), int(pw[1, 1])
            shape = (array.shape[0] + ph, array.shape[1] + pww, array.shape[2])
            src = array
            return LazyArray(shape, array.dtype, [base], lambda: _pad_now(src.t, pw, "edge"), ("rotpad", base, k, ph, pww))
    return DeviceArray(_pad_now(array.t, pw, mode))


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
    return DeviceArray(a.t.unsqueeze(int(axis)), base=a) if isinstance(axis, (int, np.integer)) else NotImplemented


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
_STAGE = {}          # size class -> [slots, next]; a slot = [pinned tensor, its numpy view, event of its last copy]; LRU order
_STAGE_LOCK = __import__("threading").Lock()      # the rings are shared by every caller of the process
_STAGE_MIN = 1 << 20
_STAGE_MAX_BYTES = 256 << 20
_STAGE_CHUNK = 4 << 20


def _size_class(n):
    """bytes -> ring size: four classes per octave (at most 25 % slack), so a folder of photographs of assorted sizes
    settles on a handful of rings instead of pinning two buffers per distinct byte count"""
    step = 1 << max(n.bit_length() - 3, 12)
    return (n + step - 1) // step * step


def upload(a):
    """contiguous numpy array -> CUDA tensor.  Arrays of 1 MB and more go through a ring of two pinned buffers per size class: a
    single-threaded `np.copyto` (0.9 ms per 25 MB) and an asynchronous copy, so the host neither waits for the device work queued
    before nor depends on the runtime's pageable path, which takes 0.55 ms per 25 MB on one host and 2.4 ms on the next
    (profiles/r04_experiments.txt); smaller ones take the plain pageable copy.  (torch's `pinned.copy_(pageable)` is not used: it
    runs on the CPU thread pool, slow and erratic in a container with fewer CPUs than the machine shows.)  Rings are kept up to
    256 MB of pinned memory in all; beyond that the least recently used ring goes."""
    torch = _torch()
    a = np.ascontiguousarray(a)
    if a.nbytes < _STAGE_MIN or a.dtype.hasobject:
        return torch.from_numpy(a).cuda()
    tdt = _torch_dtype(torch, a.dtype)                 # raises for dtypes torch does not have, before anything is staged
    cls = _size_class(a.nbytes)
    with _STAGE_LOCK:
        ring = _STAGE.pop(cls, None)
        if ring is None:
            while _STAGE and 2 * (sum(_STAGE) + cls) > _STAGE_MAX_BYTES:
                _STAGE.pop(next(iter(_STAGE)))         # least recently used first (dicts keep insertion order)
            ring = [[], 0]
            for _ in range(2):
                t = torch.empty(cls, dtype=torch.uint8).pin_memory()
                ring[0].append([t, t.numpy(), None])
        _STAGE[cls] = ring                             # (re-)inserted last = most recently used
        slot = ring[0][ring[1] & 1]
        ring[1] += 1
        if slot[2] is not None:
            slot[2].synchronize()
        src = a.reshape(-1).view(np.uint8)
        if a.nbytes >= 4 * _STAGE_CHUNK:
            # in pieces: the DMA of piece k runs under the host copy of piece k + 1 (a 25-MB frame: 0.9 ms of np.copyto + 0.45 ms of
            # DMA become ~1.0 ms)
            d8 = torch.empty(a.nbytes, dtype=torch.uint8, device="cuda")
            for lo in range(0, a.nbytes, _STAGE_CHUNK):
                hi = min(lo + _STAGE_CHUNK, a.nbytes)
                np.copyto(slot[1][lo:hi], src[lo:hi])
                d8[lo:hi].copy_(slot[0][lo:hi], non_blocking=True)
            d = d8.view(tdt).reshape(a.shape)
        else:
            np.copyto(slot[1][:a.nbytes], src)
            d = slot[0][:a.nbytes].cuda(non_blocking=True).view(tdt).reshape(a.shape)
        ev = torch.cuda.Event()
        ev.record()
        slot[2] = ev
    return d


def _torch_dtype(torch, dt):
    return torch.from_numpy(np.empty(0, dtype=dt)).dtype


def download(t):
    """torch tensor -> numpy array.  1 MB and more land in a pinned host tensor from torch's caching host allocator (the DMA runs at
    the link's rate: 0.6 ms instead of 2.9 ms for a 25 MB frame) and the array returned is a view of it -- ordinary writable
    memory to its user, handed back to the allocator's cache when the array is dropped.
    The MEMORY LAYOUT is numpy's: the result of an elementwise operation on a transposed view is laid out like the view (order
    'K'), and so is the array this returns for a permuted dense tensor -- torch's float32 reductions (the reference's mPSNR,
    common/utils.py:168-175) add in memory order, so a C-contiguous copy would change their last bits."""
    torch = _torch()
    t = t.detach()
    perm = None
    if t.dim() > 1 and not t.is_contiguous():
        order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))
        tp = t.permute(*order)
        if tp.is_contiguous():                                 # a permutation of a dense array: copy it as it lies
            t, perm = tp, [order.index(d) for d in range(len(order))]
    if not t.is_cuda:
        a = t.contiguous().numpy().copy()                      # (CPU tensors: tests only) never an alias of the tensor
    else:
        t = t.contiguous()
        if t.numel() * t.element_size() < _STAGE_MIN:
            a = t.cpu().numpy()
        else:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            a = h.numpy()
    return a.transpose(perm) if perm is not None else a
