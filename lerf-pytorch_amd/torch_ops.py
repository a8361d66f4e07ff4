"""`torch.ops.lerf.*`: dispatcher-visible wrappers over the C ABI (forward only).

LUT sets are registered once (`register_luts`) and referred to by handle, since an op schema
carries tensors and scalars only:

    h = torch_ops.register_luts(LutSet.shipped("lerf-g"))
    out = torch.ops.lerf.sr_fused(img_u8, h, 2.0, 2.0, 2, 10.0)
    feat, hq = torch.ops.lerf.lut_stages(img_u8, h)
    out = torch.ops.lerf.resize_gauss(feat_f32, rho, sx, sy, 2.0, 2.0, 2, 10.0)
    out = torch.ops.lerf.resize_linear(feat_f32, alpha, 2.0, 2.0, 1.0)
    out = torch.ops.lerf.warp_fused(img_u8, h, M_3x3_f64, 2160, 3840, 2, 10.0)
"""
from __future__ import annotations

import torch

from . import ops

_LUTS = []
_GEO = {}


def register_luts(lutset) -> int:
    _LUTS.append(lutset)
    return len(_LUTS) - 1


def _geo(hw, sh, sw, support, device):
    key = (int(hw[0]), int(hw[1]), float(sh), float(sw), int(support), str(device))
    if key not in _GEO:
        _GEO[key] = ops.SrGeometry(hw, [sh, sw], None, support, device)
    return _GEO[key]


@torch.library.custom_op("lerf::lut_stages", mutates_args=(), device_types="cuda")
def lut_stages(img: torch.Tensor, luts: int) -> tuple[torch.Tensor, torch.Tensor]:
    return ops.lut_stages(img, _LUTS[luts])


@lut_stages.register_fake
def _(img, luts):
    oC = _LUTS[luts].oC
    return torch.empty_like(img), img.new_empty(tuple(img.shape) + (oC,))


@torch.library.custom_op("lerf::sr_fused", mutates_args=(), device_types="cuda")
def sr_fused(img: torch.Tensor, luts: int, scale_h: float, scale_w: float, support: int, max_sigma: float) -> torch.Tensor:
    L = _LUTS[luts]
    linear = L.oC == 1
    geo = _geo(img.shape[-3:-1], scale_h, scale_w, 2 if linear else support, img.device)
    return ops.sr_fused_u8(img, L, geo, "linear" if linear else "gauss", 1.0 if linear else max_sigma)   # a fresh tensor


@sr_fused.register_fake
def _(img, luts, scale_h, scale_w, support, max_sigma):
    from ._lib import out_size
    shp = list(img.shape)
    shp[-3], shp[-2] = out_size(shp[-3], scale_h), out_size(shp[-2], scale_w)
    return img.new_empty(shp)


@torch.library.custom_op("lerf::resize_gauss", mutates_args=(), device_types="cuda")
def resize_gauss(feat: torch.Tensor, rho: torch.Tensor, sigma_x: torch.Tensor, sigma_y: torch.Tensor,
                 scale_h: float, scale_w: float, support: int, max_sigma: float) -> torch.Tensor:
    B, C, H, W = feat.shape
    geo = _geo((H, W), scale_h, scale_w, support, feat.device)
    r = lambda t: t.reshape(B * C, H, W)
    out = ops.resize_planar(r(feat), [r(rho), r(sigma_x), r(sigma_y)], geo, "gauss", max_sigma, out="f32")
    return out.reshape(B, C, geo.out_hw[0], geo.out_hw[1])


@resize_gauss.register_fake
def _(feat, rho, sigma_x, sigma_y, scale_h, scale_w, support, max_sigma):
    from ._lib import out_size
    B, C, H, W = feat.shape
    return feat.new_empty((B, C, out_size(H, scale_h), out_size(W, scale_w)), dtype=torch.float32)


@torch.library.custom_op("lerf::resize_linear", mutates_args=(), device_types="cuda")
def resize_linear(feat: torch.Tensor, alpha: torch.Tensor, scale_h: float, scale_w: float, max_sigma: float) -> torch.Tensor:
    B, C, H, W = feat.shape
    geo = _geo((H, W), scale_h, scale_w, 2, feat.device)
    out = ops.resize_planar(feat.reshape(B * C, H, W), [alpha.reshape(B * C, H, W)], geo, "linear", max_sigma, out="f32")
    return out.reshape(B, C, geo.out_hw[0], geo.out_hw[1])


@resize_linear.register_fake
def _(feat, alpha, scale_h, scale_w, max_sigma):
    from ._lib import out_size
    B, C, H, W = feat.shape
    return feat.new_empty((B, C, out_size(H, scale_h), out_size(W, scale_w)), dtype=torch.float32)


@torch.library.custom_op("lerf::warp_fused", mutates_args=(), device_types="cuda")
def warp_fused(img: torch.Tensor, luts: int, matrix: torch.Tensor, out_h: int, out_w: int, support: int,
               max_sigma: float) -> torch.Tensor:
    """uint8 [H,W,3] -> uint8 [out_h,out_w,3]: LUT stages (tile-fused kernel) + homographic resampling, the body of
    eltr._worker in resample/eval_lut_warp.py:100-222.  `matrix`: 3x3 float64 (input -> output coordinates)."""
    L = _LUTS[luts]
    linear = L.oC == 1
    H, W, Cn = img.shape
    geo = ops.WarpGeometry((H, W), matrix, (out_h, out_w), 2 if linear else support)
    kind, ms = ("linear", 1.0) if linear else ("gauss", max_sigma)
    if Cn == 3 and L.struct.fused_pack is not None:
        return ops.warp_packed(ops.stages_packed(img, L), geo, kind, ms, out="u8")
    feat, hq = ops.lut_stages(img, L)
    return ops.warp_hwc_u8(feat, hq, geo, kind, ms, out="u8")


@warp_fused.register_fake
def _(img, luts, matrix, out_h, out_w, support, max_sigma):
    return img.new_empty((out_h, out_w, img.shape[2]))
