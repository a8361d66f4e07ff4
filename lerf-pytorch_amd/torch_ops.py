"""`torch.ops.lerf.*`: the hot path as PyTorch dispatcher ops.

The schemas and the device kernels' launchers are registered in C++ (`csrc/lerf_torch.cpp`, TORCH_LIBRARY /
TORCH_LIBRARY_IMPL under the HIP backend's dispatch key) in `liblerf_torch.so`, which this module loads.  On top of
that it attaches what is naturally Python: the fake (meta) kernels for shape inference / torch.compile tracing and the
autograd formulas of the two float resamplers (their backward is itself an op, `lerf::resize_backward`).  There is no
CPU kernel: CPU tensors are refused by the dispatcher, and a missing library raises here.

    s1, s2, pack = torch_ops.lut_args(LutSet.shipped("lerf-g"))
    out      = torch.ops.lerf.sr_fused(img_u8, s1, s2, pack, 2.0, 2.0, 2, 10.0)
    feat, hq = torch.ops.lerf.lut_stages(img_u8, s1, s2)
    out      = torch.ops.lerf.resize_gauss(feat_f32, rho, sx, sy, 2.0, 2.0, 2, 10.0)       # differentiable
    out      = torch.ops.lerf.resize_linear(feat_f32, alpha, 2.0, 2.0, 1.0)                # differentiable
    out      = torch.ops.lerf.warp_fused(img_u8, s1, s2, pack, M_3x3_f64, 2160, 3840, 2, 10.0)
"""
from __future__ import annotations

import os

import torch

from . import _lib

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblerf_torch.so")
if not os.path.exists(LIB_PATH):
    raise _lib.LerfError("liblerf_torch.so not found at %s -- build it with `python __graft_entry__.py`" % LIB_PATH)
if os.path.abspath(_lib.LIB_PATH) != os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblerf_hip.so"):
    # liblerf_torch.so links against the product liblerf_hip.so next to it: with a variant build selected (_lib.use_library)
    # the process would hold BOTH builds and the ops would silently run the product kernels
    raise _lib.LerfError("torch_ops: a variant library is selected (%s); the dispatcher ops exist for the product build only" % _lib.LIB_PATH)
_lib.lib()                                   # liblerf_hip.so first (the op library links against it)
torch.ops.load_library(LIB_PATH)


def lut_args(lutset):
    """(luts_s1, luts_s2, pack) of a LutSet in the order the ops expect (modes "sct")."""
    if lutset.modes != "sct" or lutset.modes2 != "sct":
        raise ValueError("the dispatcher ops cover the shipped mode set 'sct'/'sct'")
    t = lutset.tensors
    s1 = [t["s1_%sr0" % m] for m in "sct"]
    s2 = [t["s2_%sr%d" % (m, r)] for m in "sct" for r in (0, 1)]
    return s1, s2, t.get("fused_pack")


def _out_hw(h, w, sh, sw):
    return _lib.out_size(h, sh), _lib.out_size(w, sw)


@torch.library.register_fake("lerf::lut_stages")
def _(img, luts_s1, luts_s2):
    oC = luts_s2[0].numel() // _lib.LERF_LUT_ENTRIES
    return torch.empty_like(img), img.new_empty(tuple(img.shape) + (oC,))


@torch.library.register_fake("lerf::sr_fused")
def _(img, luts_s1, luts_s2, pack, scale_h, scale_w, support, max_sigma):
    shp = list(img.shape)
    shp[-3], shp[-2] = _out_hw(shp[-3], shp[-2], scale_h, scale_w)
    return img.new_empty(shp)


@torch.library.register_fake("lerf::warp_fused")
def _(img, luts_s1, luts_s2, pack, matrix, out_h, out_w, support, max_sigma):
    return img.new_empty((out_h, out_w, img.shape[2]))


@torch.library.register_fake("lerf::resize_gauss")
def _(feat, rho, sigma_x, sigma_y, scale_h, scale_w, support, max_sigma):
    B, C, H, W = feat.shape
    return feat.new_empty((B, C) + _out_hw(H, W, scale_h, scale_w))


@torch.library.register_fake("lerf::resize_linear")
def _(feat, alpha, scale_h, scale_w, max_sigma):
    B, C, H, W = feat.shape
    return feat.new_empty((B, C) + _out_hw(H, W, scale_h, scale_w))


@torch.library.register_fake("lerf::resize_backward")
def _(kind, grad_out, feat, h0, h1, h2, scale_h, scale_w, support, max_sigma):
    return tuple(torch.empty_like(feat) for _ in range(4))


# ---- autograd of the float resamplers: what autograd derives for SteeringGaussianResize2dTorch.resize /
#      AmplifiedLinearResize2dTorch.resize (resize_right2d_torch.py:154-247), computed by lerf_resize_bwd_f32
def _gauss_setup(ctx, inputs, output):
    feat, rho, sx, sy, sh, sw, S, ms = inputs
    ctx.save_for_backward(feat, rho, sx, sy)
    ctx.meta = (sh, sw, S, ms)


def _gauss_backward(ctx, grad_out):
    feat, rho, sx, sy = ctx.saved_tensors
    sh, sw, S, ms = ctx.meta
    gx, g0, g1, g2 = torch.ops.lerf.resize_backward(0, grad_out.contiguous(), feat, rho, sx, sy, sh, sw, S, ms)
    return gx, g0, g1, g2, None, None, None, None


def _linear_setup(ctx, inputs, output):
    feat, alpha, sh, sw, ms = inputs
    ctx.save_for_backward(feat, alpha)
    ctx.meta = (sh, sw, ms)


def _linear_backward(ctx, grad_out):
    feat, alpha = ctx.saved_tensors
    sh, sw, ms = ctx.meta
    gx, g0, _, _ = torch.ops.lerf.resize_backward(1, grad_out.contiguous(), feat, alpha, alpha, alpha, sh, sw, 2, ms)
    return gx, g0, None, None, None


torch.library.register_autograd("lerf::resize_gauss", _gauss_backward, setup_context=_gauss_setup)
torch.library.register_autograd("lerf::resize_linear", _linear_backward, setup_context=_linear_setup)
