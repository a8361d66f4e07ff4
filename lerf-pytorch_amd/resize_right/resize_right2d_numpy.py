"""numpy-facing resampler classes: drop-in for the LeRF classes of the
reference's resize_right/resize_right2d_numpy.py (same names, constructor
arguments, set_shape/resize/warp signatures, float64 [C,oH,oW] results).

Inputs are numpy arrays; they are staged to the MI355X, resampled by the HIP
kernels in float64 arithmetic (LERF_F64 outputs) and copied back.  There is no
CPU implementation here.
"""
from __future__ import annotations

from math import ceil

import numpy as np

from .. import _lib, ops


def _to_dev(a):
    """numpy array -> float32 device tensor; a lazy.DeviceArray (a result that never left the device) is used in place"""
    torch = _lib.require_gpu()
    from .. import lazy
    if isinstance(a, lazy.DeviceArray):
        return a.t.to(torch.float32)
    return lazy.upload(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))


def _result(out, inputs):
    """float64 device result -> what the caller gets: a device-backed array when the inputs were device-backed (or lazy
    results are enabled), else the numpy array of the reference's contract"""
    from .. import lazy
    if lazy.enabled() or any(isinstance(a, lazy.DeviceArray) for a in inputs):
        return lazy.DeviceArray(out)
    return out.cpu().numpy()


class Resize2dNumpy(object):
    """Geometry holder (reference: resize_right2d_numpy.py:10-140)."""
    _scales_distances = False

    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        self._pad_code = _lib.pad_mode_code(pad_mode, _lib.NUMPY_PAD_MODES)     # np.pad(input, ..., mode=pad_mode) (:208, :560)
        self.eps = np.finfo(np.float32).eps
        self.device = device
        self.support_sz = support_sz
        self.pad_mode = pad_mode
        self.antialias = False

    def set_shape(self, in_shape, scale_factors=None, out_shape=None):
        in_shape = list(in_shape)                       # [C, H, W]
        if len(in_shape) != 3:
            raise ValueError("in_shape must be [C, H, W]")
        out_hw = None
        if out_shape is not None:                       # :26-31
            out_shape = list(out_shape) + list(in_shape[len(out_shape):])
            out_hw = (out_shape[1], out_shape[2])
            if scale_factors is None:
                scale_factors = [o / i for o, i in zip(out_shape, in_shape)][1:]
        if scale_factors is None:
            raise ValueError("either scale_factors or out_shape is required")
        if not isinstance(scale_factors, (list, tuple)):
            scale_factors = [scale_factors, scale_factors]
        scale_factors = [1] * (3 - len(scale_factors)) + list(scale_factors)
        self.in_shape = in_shape
        self.scale_factors = [float(s) for s in scale_factors]
        # anti-aliasing for down-sampling (:51-55).  The reference indexes its [C, H, W] scale list with [0] and [1],
        # i.e. it looks at the channel factor (always 1) and the ROW factor only: a column-only down-sampling gets no
        # anti-aliasing and the enlargement is ceil(S / sh).  Reproduced.  Like the reference, the enlarged support
        # and the flag stay on the object: a later set_shape never resets them.
        if self.scale_factors[0] < 1.0 or self.scale_factors[1] < 1.0:
            self.antialias = True
            self.min_scale_factor = min([self.scale_factors[1], self.scale_factors[0]])
            self.support_sz = ceil(self.support_sz / self.min_scale_factor)
        dis_scale = self.min_scale_factor if (self.antialias and self._scales_distances) else 1.0
        self.geo = ops.SrGeometry(in_shape[1:], self.scale_factors[1:], out_hw, self.support_sz, dis_scale=dis_scale,
                                  pad_mode=self._pad_code)
        self.out_shape = [ceil(self.scale_factors[0] * in_shape[0]), self.geo.out_hw[0], self.geo.out_hw[1]]
        self.in_sz = [in_shape[1], in_shape[2]]
        self.out_sz = [self.geo.out_hw[0], self.geo.out_hw[1]]
        self.pad_vec = self.geo.pad_vec

    # The dense geometry attributes get_distance leaves on the reference object (:106-140).  The kernels work from the
    # two 1-D tables per axis; the dense maps ([oH*S, oW*S] int64 / [C, oH*S, oW*S] float64: 796 MB each at 1080p -> 4K)
    # are materialised on access only.  Patch enumeration: numpy meshgrid 'xy' (:95-98) -- inside a patch the row
    # offset varies along the COLUMN index.
    def _dense(self):
        h, S = self.geo.host, self.support_sz
        oH, oW = self.out_sz
        k = np.arange(S)
        pr, pc = self.pad_vec[1][0], self.pad_vec[2][0]
        fx = np.repeat(h["left_r"].astype(np.int64) + pr, S)[:, None] + np.tile(k, oW)[None, :]
        fy = np.repeat(h["left_c"].astype(np.int64) + pc, S)[None, :] + np.tile(k, oH)[:, None]
        dx = np.tile(np.repeat(h["dis_r"].reshape(-1, S), S, axis=0), (1, oW))
        dy = np.tile(np.repeat(h["dis_c"].reshape(-1, S).T, S, axis=1), (oH, 1))
        C0 = self.in_shape[0]
        return fx, fy, np.broadcast_to(dx[None], (C0,) + dx.shape).copy(), np.broadcast_to(dy[None], (C0,) + dy.shape).copy()

    field_of_view_x = property(lambda self: self._dense()[0])
    field_of_view_y = property(lambda self: self._dense()[1])
    dis_x = property(lambda self: self._dense()[2])
    dis_y = property(lambda self: self._dense()[3])

    def _run(self, kind, input, hypers, max_sigma):
        x = _to_dev(input)
        if list(x.shape) != list(self.in_shape):
            raise ValueError("input shape {} does not match set_shape({})".format(list(x.shape), self.in_shape))
        hs = [_to_dev(h) for h in hypers]
        from .. import lazy
        ins = [input] + list(hypers)
        tags = [a.exact_u8() if isinstance(a, lazy.DeviceArray) else None for a in ins]
        if self._pad_code == 0 and kind in ("gauss", "linear") and all(t is not None for t in tags) and tags[0][1] == 1.0 \
                and all(t[1] == 255.0 for t in tags[1:]):
            # every operand is a stage output known exactly as uint8 (feat) / uint8 numerators over 255 (hyper): deferred
            # (lazy.LazyArray) -- the caller's `np.clip(np.round(out).transpose((1, 2, 0)), 0, 255).astype(np.uint8)`
            # (resample/eval_lut_sr.py:663-665) then runs the resampler ONCE on the uint8 maps with uint8 HWC output (float32
            # production arithmetic + the float64 tie guard: the same bytes) instead of 199 MB of float64 and three more passes
            # over them; any other use of the result: the float64 kernel, as before
            geo, f8, h8 = self.geo, tags[0][0], [t[0] for t in tags[1:]]
            return lazy.LazyArray((x.shape[0], geo.out_hw[0], geo.out_hw[1]), np.float64, [a for a in ins if isinstance(a, lazy.DeviceArray)],
                                  lambda: ops.resize_planar(x, hs, geo, kind, max_sigma, out="f64"),
                                  ("resize", lambda: ops.resize_planar_u8(f8, h8, geo, kind, max_sigma)))
        out = ops.resize_planar(x, hs, self.geo, kind, max_sigma, out="f64")
        return _result(out, ins)


class SteeringGaussianResize2dNumpy(Resize2dNumpy):
    _scales_distances = True        # weights = m * sk_weight(..., m * dis_x, m * dis_y) when anti-aliasing (:186-193)

    def __init__(self, support_sz=4, device="CPU", pad_mode="constant", max_sigma=10):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def resize(self, input, rho, sigma_x, sigma_y):
        return self._run("gauss", input, [rho, sigma_x, sigma_y], self.max_sigma)


class AmplifiedLinearResize2dNumpy(Resize2dNumpy):
    def __init__(self, support_sz=2, device="CPU", pad_mode="constant", max_sigma=1):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def resize(self, input, alpha):
        return self._run("linear", input, [alpha], self.max_sigma)


class Warp2dNumpy(object):
    """Homography geometry holder (reference: resize_right2d_numpy.py:284-407)."""
    kind = None

    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        self._pad_code = _lib.pad_mode_code(pad_mode, _lib.NUMPY_PAD_MODES)     # np.pad(input, ..., mode=pad_mode) (:208, :560)
        self.eps = np.finfo(np.float32).eps
        self.device = device
        self.support_sz = support_sz
        self.pad_mode = pad_mode
        self.antialias = False

    def set_shape(self, in_shape, matrix, out_shape):
        in_shape = list(in_shape)
        out_shape = list(out_shape) + list(in_shape[len(out_shape):])      # :301
        self.in_shape, self.out_shape, self.matrix = in_shape, out_shape, matrix
        self.in_sz = [in_shape[1], in_shape[2]]
        self.out_sz = [out_shape[1], out_shape[2]]
        self.geo = ops.WarpGeometry(self.in_sz, matrix, self.out_sz, self.support_sz, pad_mode=self._pad_code)
        self.pad_vec = self.geo.pad_vec

    def _run(self, kind, input, hypers, max_sigma):
        x = _to_dev(input)
        if list(x.shape) != list(self.in_shape):
            raise ValueError("input shape {} does not match set_shape({})".format(list(x.shape), self.in_shape))
        hs = [_to_dev(h) for h in hypers]
        return _result(ops.warp_planar(x, hs, self.geo, kind, max_sigma, out="f64"), [input] + list(hypers))


class NearestWarp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=1, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("nearest", input, [], 1.0)


class SteeringGaussianWarp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=4, device="CPU", pad_mode="constant", max_sigma=10):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def warp(self, input, rho, sigma_x, sigma_y):
        return self._run("gauss", input, [rho, sigma_x, sigma_y], self.max_sigma)


class AmplifiedLinearWarp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=2, device="CPU", pad_mode="constant", max_sigma=1):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def warp(self, input, alpha):
        return self._run("linear", input, [alpha], self.max_sigma)


# fixed-kernel baselines of the reference (resize_right2d_numpy.py:451-494, resize_right2d_torch.py:372-388);
# weights from resize_right/interp_methods.py evaluated in float64 on the device
class BicubicWarp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("cubic", input, [], 1.0)


class BilinearWarp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=2, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("bilinear", input, [], 1.0)


class Lanczos2Warp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("lanczos2", input, [], 1.0)


class Lanczos3Warp2dNumpy(Warp2dNumpy):
    def __init__(self, support_sz=6, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("lanczos3", input, [], 1.0)
