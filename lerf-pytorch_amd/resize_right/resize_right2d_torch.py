"""torch-facing resampler classes: drop-in for the LeRF classes of the
reference's resize_right/resize_right2d_torch.py ([B,C,H,W] tensors on the GPU).

SR geometry follows the torch classes' own float32 arithmetic (resize_right2d_torch.py:48-103, bit-equal tables:
lerf_sr_axis_tables_f32), warp geometry the float64 one; SR returns float32, warps return float64 like
the reference (its warp distances are double, resize_right2d_torch.py:286-296).
The SR classes carry autograd (HIP backward, `_ResizeFn`); the warp classes are forward only.
"""
from __future__ import annotations

from math import ceil

import torch

from .. import _lib, ops


class _ResizeFn(torch.autograd.Function):
    """lerf_resize forward, lerf_resize_bwd_f32 backward (the gradient autograd derives for
    resize_right2d_torch.py:154-247) -- used when an input of resize() requires grad (train_model.py:431-441)."""

    @staticmethod
    def forward(ctx, geo, kind, max_sigma, x, *hs):
        x = x.detach().contiguous().float()
        hs = [h.detach().contiguous().float() for h in hs]
        out = ops.resize_planar(x, hs, geo, kind, max_sigma, out="f32")
        ctx.save_for_backward(x, *hs)
        ctx.meta = (geo, kind, float(max_sigma))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import ctypes as C
        x, *hs = ctx.saved_tensors
        geo, kind, max_sigma = ctx.meta
        g = grad_out.contiguous().float()
        N, H, W = x.shape
        need = ctx.needs_input_grad[3:]
        grads = [torch.zeros_like(x) if need[k] else None for k in range(1 + len(hs))]
        hp = [C.c_void_p(h.data_ptr()) for h in hs] + [C.c_void_p(None)] * (3 - len(hs))
        gp = [C.c_void_p(t.data_ptr() if t is not None else None) for t in grads] + [C.c_void_p(None)] * (3 - len(hs))
        _lib.check(_lib.lib().lerf_resize_bwd_f32(C.c_void_p(x.data_ptr()), hp[0], hp[1], hp[2], N, H, W, geo.ref(),
                                                  _lib.KINDS[kind], max_sigma, C.c_void_p(g.data_ptr()), gp[0], gp[1], gp[2],
                                                  gp[3], _lib.current_stream()), "lerf_resize_bwd_f32")
        return (None, None, None) + tuple(grads)


def _check_dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("{} must be a tensor on the GPU (there is no CPU path)".format(name))


class Resize2dTorch(object):
    def __init__(self, support_sz=4, device="GPU", pad_mode="constant"):
        self._pad_code = _lib.pad_mode_code(pad_mode, _lib.TORCH_PAD_MODES)     # F.pad(input, ..., mode=pad_mode) (:189, :362)
        self.eps = torch.finfo(torch.float32).eps
        self.device = device
        self.init_support_sz = support_sz
        self.pad_mode = pad_mode
        self.antialias = False

    def set_shape(self, in_shape, scale_factors=None, out_shape=None):
        self.support_sz = self.init_support_sz                 # :19
        in_shape = list(in_shape)
        if len(in_shape) != 4:
            raise ValueError("in_shape must be [B, C, H, W]")
        out_hw = None
        if out_shape is not None:                              # :26-29
            out_shape = list(in_shape[:-len(out_shape)]) + list(out_shape)
            out_hw = (out_shape[2], out_shape[3])
            if scale_factors is None:
                scale_factors = [o / i for o, i in zip(out_shape, in_shape)][2:]
        if scale_factors is None:
            raise ValueError("either scale_factors or out_shape is required")
        if not isinstance(scale_factors, (list, tuple)):
            scale_factors = [scale_factors, scale_factors]
        scale_factors = [1] * (4 - len(scale_factors)) + list(scale_factors)
        self.in_shape = in_shape
        self.scale_factors = [float(s) for s in scale_factors]
        self.geo = ops.SrGeometry(in_shape[2:], self.scale_factors[2:], out_hw, self.support_sz, arithmetic="torch32",
                                  pad_mode=self._pad_code)
        self.out_shape = [ceil(self.scale_factors[0] * in_shape[0]), ceil(self.scale_factors[1] * in_shape[1]),
                          self.geo.out_hw[0], self.geo.out_hw[1]]
        pr, pc = self.geo.pad_vec[1], self.geo.pad_vec[2]
        self.pad_vec = [pr[0], pr[1], pc[0], pc[1]]            # the reference's ordering (:93-95)
        if tuple(pr) != tuple(pc):
            # F.pad consumes pad_vec last-dimension first, so the reference pads the columns with the ROW pads and
            # vice versa (:189); harmless while both are (S/2, S/2) -- every up-sampling -- but a shifted / out-of-range
            # gather when they differ (some down-samplings).  That case is not reproduced.
            raise NotImplementedError("row pads {} != column pads {}: the reference mis-pads this geometry".format(pr, pc))

    # dense geometry attributes of the reference object (:48-103), materialised on access only; torch.meshgrid 'ij'
    # (:72-76): inside a patch the row offset varies along the ROW index; dis_* are [B, 1, oH*S, oW*S] float32
    def _dense(self):
        import numpy as np
        h, S = self.geo.host, self.support_sz
        oH, oW = self.geo.out_hw
        k = np.arange(S)
        pr, pc = self.geo.pad_vec[1][0], self.geo.pad_vec[2][0]
        fx = (np.repeat(h["left_r"].astype(np.int64) + pr, S) + np.tile(k, oH))[:, None] + np.zeros((1, oW * S), np.int64)
        fy = (np.repeat(h["left_c"].astype(np.int64) + pc, S) + np.tile(k, oW))[None, :] + np.zeros((oH * S, 1), np.int64)
        dx = np.repeat(h["dis_r32"].reshape(-1)[:, None], oW * S, axis=1)
        dy = np.repeat(h["dis_c32"].reshape(-1)[None, :], oH * S, axis=0)
        B = self.in_shape[0]
        dev = self.geo.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return t(fx), t(fy), t(dx)[None, None].repeat(B, 1, 1, 1), t(dy)[None, None].repeat(B, 1, 1, 1)

    field_of_view_x = property(lambda self: self._dense()[0])
    field_of_view_y = property(lambda self: self._dense()[1])
    dis_x = property(lambda self: self._dense()[2])
    dis_y = property(lambda self: self._dense()[3])

    def _run(self, kind, input, hypers, max_sigma):
        _check_dev(input, "input")
        B, Cn, H, W = input.shape
        if [H, W] != list(self.in_shape[2:]):
            raise ValueError("input shape does not match set_shape")
        x = input.reshape(B * Cn, H, W)
        hs = []
        for h in hypers:
            _check_dev(h, "hyper-parameter map")
            hs.append(h.reshape(B * Cn, H, W))
        if kind in ("gauss", "linear") and torch.is_grad_enabled() and any(t.requires_grad for t in [x] + hs):
            if self._pad_code != 0:
                raise NotImplementedError("autograd is implemented for pad_mode='constant' (what train_model.py uses)")
            out = _ResizeFn.apply(self.geo, kind, max_sigma, x, *hs)
        else:
            out = ops.resize_planar(x, hs, self.geo, kind, max_sigma, out="f32")
        return out.reshape(B, Cn, self.geo.out_hw[0], self.geo.out_hw[1])


    kind = None                                                # fixed-kernel subclasses name their interp_methods kernel

    def resize(self, input):
        """default interpolation process (:105-129): weight(dis_x, dis_y) normalised over the patch."""
        if self.kind is None:
            raise NotImplementedError("Resize2dTorch.resize needs a subclass with a weight kernel")
        return self._run(self.kind, input, [], 1.0)


class BicubicResize2dTorch(Resize2dTorch):                     # :131-138, interp_methods.cubic2d
    kind = "cubic"

    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)


class BilinearResize2dTorch(Resize2dTorch):                    # interp_methods.linear2d on the same base class
    kind = "bilinear"

    def __init__(self, support_sz=2, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)


class Lanczos2Resize2dTorch(Resize2dTorch):                    # interp_methods.lanczos2d
    kind = "lanczos2"

    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)


class Lanczos3Resize2dTorch(Resize2dTorch):                    # interp_methods.lanczos3d
    kind = "lanczos3"

    def __init__(self, support_sz=6, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)


class SteeringGaussianResize2dTorch(Resize2dTorch):
    def __init__(self, support_sz=4, device="GPU", pad_mode="constant", max_sigma=10):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def resize(self, input, rho, sigma_x, sigma_y):
        return self._run("gauss", input, [rho, sigma_x, sigma_y], self.max_sigma)


class AmplifiedLinearResize2dTorch(Resize2dTorch):
    def __init__(self, support_sz=2, device="GPU", pad_mode="constant", max_sigma=1):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def resize(self, input, alpha):
        return self._run("linear", input, [alpha], self.max_sigma)


class Warp2dTorch(object):
    def __init__(self, support_sz=4, device="GPU", pad_mode="constant"):
        self._pad_code = _lib.pad_mode_code(pad_mode, _lib.TORCH_PAD_MODES)     # F.pad(input, ..., mode=pad_mode) (:189, :362)
        self.eps = torch.finfo(torch.float32).eps
        self.device = device
        self.support_sz = support_sz
        self.pad_mode = pad_mode

    def set_shape(self, in_shape, matrix, out_shape):
        in_shape = list(in_shape)
        out_shape = list(out_shape) + list(in_shape[len(out_shape):])      # :263
        self.in_shape, self.out_shape, self.matrix = in_shape, out_shape, matrix
        self.in_sz = [in_shape[2], in_shape[3]]
        self.out_sz = [out_shape[2], out_shape[3]]
        self.geo = ops.WarpGeometry(self.in_sz, matrix, self.out_sz, self.support_sz, pad_mode=self._pad_code)
        pr, pc = self.geo.pad_vec[1], self.geo.pad_vec[2]
        self.pad_vec = [pc[0], pc[1], pr[0], pr[1]]                         # last dim first (:330-332)

    def _run(self, kind, input, hypers, max_sigma):
        _check_dev(input, "input")
        B, Cn, H, W = input.shape
        if [H, W] != list(self.in_sz):
            raise ValueError("input shape does not match set_shape")
        x = input.reshape(B * Cn, H, W)
        hs = []
        for h in hypers:
            _check_dev(h, "hyper-parameter map")
            hs.append(h.reshape(B * Cn, H, W))
        out = ops.warp_planar(x, hs, self.geo, kind, max_sigma, out="f64")
        return out.reshape(B, Cn, self.out_sz[0], self.out_sz[1])


class NearestWarp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=1, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("nearest", input, [], 1.0)


class SteeringGaussianWarp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=4, device="GPU", pad_mode="constant", max_sigma=10):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def warp(self, input, rho, sigma_x, sigma_y):
        return self._run("gauss", input, [rho, sigma_x, sigma_y], self.max_sigma)


class AmplifiedLinearWarp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=2, device="GPU", pad_mode="constant", max_sigma=1):
        super().__init__(support_sz, device, pad_mode)
        self.max_sigma = max_sigma

    def warp(self, input, alpha):
        return self._run("linear", input, [alpha], self.max_sigma)


# fixed-kernel baselines of the reference (resize_right2d_numpy.py:451-494, resize_right2d_torch.py:372-388);
# weights from resize_right/interp_methods.py evaluated in float64 on the device
class BicubicWarp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("cubic", input, [], 1.0)


class BilinearWarp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=2, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("bilinear", input, [], 1.0)


class Lanczos2Warp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=4, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("lanczos2", input, [], 1.0)


class Lanczos3Warp2dTorch(Warp2dTorch):
    def __init__(self, support_sz=6, device="CPU", pad_mode="constant"):
        super().__init__(support_sz, device, pad_mode)

    def warp(self, input):
        return self._run("lanczos3", input, [], 1.0)
