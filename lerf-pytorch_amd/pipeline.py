"""End-to-end LUT pipelines: the counterpart of `eltr._worker` in the
reference's resample/eval_lut_sr.py:514-665 (SR) and
resample/eval_lut_warp.py:70-222 (homographic warp), uint8 in -> uint8 out,
everything between staying on the GPU.
"""
from __future__ import annotations

import numpy as np

from . import _lib, ops
from .luts import LutSet


class LerfEngine:
    """One model's LUTs resident in HBM + cached geometry per shape.

    support / max_sigma follow common/option.py:25,29 (suppSize=2, maxSigma=10);
    LeRF-L (`linear=True`) always runs S=2, max_sigma=1 because the reference
    harness builds AmplifiedLinearResize2dNumpy() with its defaults
    (eval_lut_sr.py:482-484, eval_lut_warp.py:37-39).
    """

    def __init__(self, luts: LutSet, support=2, max_sigma=10.0):
        self.luts = luts
        self.linear = luts.oC == 1
        self.kind = "linear" if self.linear else "gauss"
        self.support = 2 if self.linear else int(support)
        self.max_sigma = 1.0 if self.linear else float(max_sigma)
        self._sr_geo = {}
        # warp(): True = the tile-fused kernel where it applies (no packed maps in HBM: traffic 1/3, but 15 % more time per batch and a
        # host pass over the output per homography); False (default) = stages + packed-map warp, three launches
        self.fused_warp = False

    @classmethod
    def shipped(cls, name="lerf-g", **kw):
        return cls(LutSet.shipped(name), **kw)

    # -- helpers
    def _dev(self, img):
        torch = _lib.require_gpu()
        if isinstance(img, torch.Tensor):
            return img.to(self.luts.device), False
        a = np.ascontiguousarray(np.asarray(img))
        if a.dtype != np.uint8:
            raise ValueError("images are uint8 HWC")
        return torch.from_numpy(a).to(self.luts.device), True

    def sr_geometry(self, in_hw, scale):
        if not isinstance(scale, (list, tuple)):
            scale = (scale, scale)
        key = (int(in_hw[0]), int(in_hw[1]), float(scale[0]), float(scale[1]), self.support)
        if key not in self._sr_geo:
            self._sr_geo[key] = ops.SrGeometry(in_hw, list(scale), None, self.support, self.luts.device)
        return self._sr_geo[key]

    # -- stages 1+2
    def _fused_stages_ok(self, x):
        return x.shape[-1] in (1, 3, 4) and self.luts.struct.fused_pack is not None

    def stages(self, img):
        x, as_np = self._dev(img)
        if self._fused_stages_ok(x):
            feat, hq = ops.unpack_stages(ops.stages_packed(x, self.luts), self.luts.oC)
        else:
            feat, hq = ops.lut_stages(x, self.luts)
        return (feat.cpu().numpy(), hq.cpu().numpy()) if as_np else (feat, hq)

    # -- SR
    def sr(self, img, scale, fused=True):
        """uint8 [H,W,C] (or [N,H,W,C]) -> uint8 [ceil(sh*H), ceil(sw*W), C]."""
        x, as_np = self._dev(img)
        hw = x.shape[-3:-1]
        geo = self.sr_geometry(hw, scale)
        if fused:
            out = ops.sr_fused_u8(x, self.luts, geo, self.kind, self.max_sigma)
        else:
            if x.dim() != 3:
                raise ValueError("unfused path takes one frame")
            feat, hq = ops.lut_stages(x, self.luts)
            out = ops.resize_hwc_u8(feat, hq, geo, self.kind, self.max_sigma, out="u8")
        return out.cpu().numpy() if as_np else out

    def sr_many(self, imgs, scales):
        """Frames of DIFFERENT sizes and scale factors in one ragged launch pair (ops.sr_fused_ragged_u8): what eltr.run
        does image by image over a benchmark folder (eval_lut_sr.py:489-512).  imgs: list of uint8 [H_i,W_i,C];
        scales: one scale (or (sh, sw)) per image.  Returns a list of uint8 [oH_i,oW_i,C] (numpy in -> numpy out)."""
        xs, as_np = zip(*[self._dev(i) for i in imgs])
        geos = [self.sr_geometry(x.shape[:2], s) for x, s in zip(xs, scales)]
        outs = ops.sr_fused_ragged_u8(list(xs), self.luts, geos, self.kind, self.max_sigma)
        return [o.cpu().numpy() if n else o for o, n in zip(outs, as_np)]

    def sr_float(self, img, scale):
        """float32 [oH,oW,C] before the final rounding (for tolerance checks)."""
        x, as_np = self._dev(img)
        geo = self.sr_geometry(x.shape[:2], scale)
        feat, hq = ops.lut_stages(x, self.luts)
        out = ops.resize_hwc_u8(feat, hq, geo, self.kind, self.max_sigma, out="f32")
        return out.cpu().numpy() if as_np else out

    # -- homographic warp
    def warp(self, img, matrix, out_hw, border=4, return_mask=True, out="u8"):
        """uint8 [H,W,C] -> (uint8 [oH,oW,C], bool mask [oH,oW,C]).

        mask = validity mask of eval_lut_warp.py:197-204,229 (nearest warp of a
        white frame with a `border`-px black rim, == 255)."""
        torch = _lib.require_gpu()
        x, as_np = self._dev(img)
        H, W, Cn = x.shape
        geo = ops.WarpGeometry((H, W), matrix, out_hw, self.support)
        if out == "u8" and self.fused_warp and self._fused_stages_ok(x) and ops.warp_fused_supported(x, self.luts, geo, self.kind, self.max_sigma):
            o = ops.warp_fused_u8(x, self.luts, geo, self.kind, self.max_sigma)        # tile-fused: no packed maps in HBM
        elif self._fused_stages_ok(x) and out == "u8":
            o = ops.warp_packed(ops.stages_packed(x, self.luts), geo, self.kind, self.max_sigma, out=out)
        else:
            # float outputs: float64 arithmetic in the direct kernel (f32 = rounded once at the store)
            if self._fused_stages_ok(x):
                feat, hq = ops.unpack_stages(ops.stages_packed(x, self.luts), self.luts.oC)
            else:
                feat, hq = ops.lut_stages(x, self.luts)
            o = ops.warp_hwc_u8(feat, hq, geo, self.kind, self.max_sigma, out=out)
        mask = None
        if return_mask:
            white = torch.zeros((H, W, Cn), dtype=torch.uint8, device=x.device)
            white[border:H - border, border:W - border] = 255
            ngeo = ops.WarpGeometry((H, W), matrix, out_hw, 1)
            mask = ops.warp_hwc_u8(white, None, ngeo, "nearest", 1.0, out="f32") == 255
        if as_np:
            return o.cpu().numpy(), (mask.cpu().numpy() if mask is not None else None)
        return o, mask


    def warp_many(self, imgs, matrices, out_hws, border=4):
        """The warp harness over a folder (eval_lut_warp.py:42-68): the LUT stages of ALL images in one ragged launch pair,
        then one warp + one mask launch per image (every image has its own homography and size).
        Returns [(uint8 [oH,oW,C], bool mask)] like warp()."""
        torch = _lib.require_gpu()
        xs, as_np = zip(*[self._dev(i) for i in imgs])
        if not all(self._fused_stages_ok(x) for x in xs):
            return [self.warp(x, M, hw, border) for x, M, hw in zip(xs, matrices, out_hws)]
        packed = ops.stages_packed_ragged(list(xs), self.luts)
        res = []
        for x, pk, M, hw, n in zip(xs, packed, matrices, out_hws, as_np):
            H, W, Cn = x.shape
            o = ops.warp_packed(pk, ops.WarpGeometry((H, W), M, hw, self.support), self.kind, self.max_sigma, out="u8")
            white = torch.zeros((H, W, Cn), dtype=torch.uint8, device=x.device)
            white[border:H - border, border:W - border] = 255
            mask = ops.warp_hwc_u8(white, None, ops.WarpGeometry((H, W), M, hw, 1), "nearest", 1.0, out="f32") == 255
            res.append((o.cpu().numpy(), mask.cpu().numpy()) if n else (o, mask))
        return res


_ENGINES = {}


def _engine(model, support, max_sigma):
    key = (model, support, max_sigma)
    if key not in _ENGINES:
        _ENGINES[key] = LerfEngine.shipped(model, support=support, max_sigma=max_sigma)
    return _ENGINES[key]


def sr(img_u8_hwc, scale, model="lerf-g", support=2, max_sigma=10.0):
    return _engine(model, support, max_sigma).sr(img_u8_hwc, scale)


def warp(img_u8_hwc, matrix, out_hw, model="lerf-g", support=2, max_sigma=10.0):
    return _engine(model, support, max_sigma).warp(img_u8_hwc, matrix, out_hw)
