"""Device operators: thin wrappers that hand torch-owned HBM buffers to the
C ABI on the current HIP stream.  torch is plumbing here (memory, streams);
all arithmetic happens in liblerf_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KINDS


def _torch():
    return _lib.require_gpu()


# --------------------------------------------------------------------------- geometry
class SrGeometry:
    """Separable SR geometry (Resize2dNumpy.set_shape, resize_right2d_numpy.py:18-140)
    as two 1-D tables per axis, resident on the device."""

    def __init__(self, in_hw, scale_factors=None, out_hw=None, support=2, device=None, arithmetic="f64", dis_scale=1.0,
                 pad_mode=0):
        """arithmetic: "f64" = the numpy classes' float64 tables (normative for the eval path); "torch32" = the
        float32 tables of the reference's torch classes (resize_right2d_torch.py:48-103), bit-equal to theirs.
        dis_scale: factor applied to the distances the weights see -- the anti-aliasing of the numpy Gaussian class
        for down-sampling (`min_scale_factor * dis`, resize_right2d_numpy.py:186-193); `support` is then the enlarged
        ceil(support / min_scale_factor) of :51-55 (the caller computes it, like the reference's set_scale_and_out_sz)."""
        torch = _torch()
        H, W = int(in_hw[0]), int(in_hw[1])
        if out_hw is not None and scale_factors is None:
            scale_factors = [out_hw[0] / H, out_hw[1] / W]                 # :28-31
        if not isinstance(scale_factors, (list, tuple)):
            scale_factors = [scale_factors, scale_factors]                 # :33-37
        sh, sw = float(scale_factors[0]), float(scale_factors[1])
        if not (sh > 0.0 and sw > 0.0):
            raise ValueError("scale factors must be positive")
        if not 1 <= int(support) <= _lib.LERF_MAX_SUPPORT:
            raise NotImplementedError("support size {} (after anti-aliasing enlargement) exceeds the kernels' maximum of {}"
                                      .format(support, _lib.LERF_MAX_SUPPORT))
        if out_hw is None:
            out_hw = (_lib.out_size(H, sh), _lib.out_size(W, sw))          # :41-45
        self.in_hw, self.out_hw, self.scales, self.S = (H, W), (int(out_hw[0]), int(out_hw[1])), (sh, sw), int(support)
        self.device = torch.device(device if device is not None else "cuda")
        self.pad_mode = int(pad_mode)                                        # LERF_PAD_* of the image operand (:208)
        tables = {"f64": _lib.sr_axis_tables, "torch32": _lib.sr_axis_tables_f32}[arithmetic]
        lr, dr64, dr32, pr = tables(H, self.out_hw[0], sh, self.S)
        lc, dc64, dc32, pc = tables(W, self.out_hw[1], sw, self.S)
        if float(dis_scale) != 1.0:
            dr64, dc64 = float(dis_scale) * dr64, float(dis_scale) * dc64
            dr32, dc32 = dr64.astype(np.float32), dc64.astype(np.float32)
        self.pad_vec = ((0, 0), pr, pc)                                     # :129
        self.host = dict(left_r=lr, dis_r=dr64, dis_r32=dr32, left_c=lc, dis_c=dc64, dis_c32=dc32)
        self._upload()

    def _upload(self):
        torch = _torch()
        h = self.host
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.t = dict(left_r=up(h["left_r"]), dis_r=up(h["dis_r32"]), left_c=up(h["left_c"]), dis_c=up(h["dis_c32"]),
                      dis_r64=up(h["dis_r"]), dis_c64=up(h["dis_c"]))
        g = _lib.SrGeo()
        g.S, g.out_h, g.out_w = self.S, self.out_hw[0], self.out_hw[1]
        g.pad_mode = getattr(self, "pad_mode", 0)
        for k in ("left_r", "dis_r", "left_c", "dis_c", "dis_r64", "dis_c64"):
            setattr(g, k, self.t[k].data_ptr())
        self.struct = g

    def row_slice(self, lr_row0, lr_rows, out_row0, out_row1):
        """Geometry of a horizontal strip: the LR rows [lr_row0, lr_row0 + lr_rows) held locally
        (owned rows plus halo) produce the global output rows [out_row0, out_row1).  The row tables
        are the global ones, rebased to the strip -- the kernels never see the strip as a frame of
        its own, so non-integer scales partition exactly like integer ones."""
        g = object.__new__(SrGeometry)
        g.in_hw = (int(lr_rows), self.in_hw[1])
        g.out_hw = (int(out_row1 - out_row0), self.out_hw[1])
        g.scales, g.S, g.device, g.pad_vec = self.scales, self.S, self.device, self.pad_vec
        g.pad_mode = self.pad_mode
        h = self.host
        g.host = dict(left_r=(h["left_r"][out_row0:out_row1] - lr_row0).astype(np.int32),
                      dis_r=h["dis_r"][out_row0:out_row1], dis_r32=h["dis_r32"][out_row0:out_row1],
                      left_c=h["left_c"], dis_c=h["dis_c"], dis_c32=h["dis_c32"])
        g._upload()
        return g

    def ref(self):
        return C.byref(self.struct)


class WarpGeometry:
    """Homography geometry (Warp2dNumpy.set_shape, resize_right2d_numpy.py:292-407)."""

    def __init__(self, in_hw, matrix, out_hw, support=2, pad_mode=0):
        m = np.asarray(matrix.detach().cpu().numpy() if hasattr(matrix, "detach") else matrix, dtype=np.float64)
        if m.shape != (3, 3):
            raise ValueError("matrix must be 3x3")
        self.matrix = m
        self.minv = np.linalg.inv(m)                                       # :327
        self.in_hw, self.out_hw, self.S = (int(in_hw[0]), int(in_hw[1])), (int(out_hw[0]), int(out_hw[1])), int(support)
        pads = _lib.warp_pads(self.minv, self.in_hw, self.out_hw, self.S)
        self.pad_vec = ((0, 0), (pads[0], pads[1]), (pads[2], pads[3]))    # :392
        g = _lib.WarpGeo()
        g.S, g.out_h, g.out_w = self.S, self.out_hw[0], self.out_hw[1]
        for i, v in enumerate(self.minv.reshape(9)):
            g.minv[i] = float(v)
        g.pad_r_lo, g.pad_r_hi, g.pad_c_lo, g.pad_c_hi = pads
        g.pad_mode = self.pad_mode = int(pad_mode)                          # LERF_PAD_* of the image operand (:560)
        self.struct = g

    def ref(self):
        return C.byref(self.struct)


# --------------------------------------------------------------------------- helpers
def _planes_chw(t):
    """[N,H,W] contiguous-ish tensor -> Plane with channel = leading dim."""
    return _lib.plane(t, t.stride(1), t.stride(2), t.stride(0))


def _planes_hwc(t):
    return _lib.plane(t, t.stride(0), t.stride(1), t.stride(2))


def _out_dtype(name):
    torch = _torch()
    return {"u8": torch.uint8, "f32": torch.float32, "f64": torch.float64}[name]


# --------------------------------------------------------------------------- A1
def lut_interp_i16(img_u8_chw, h, w, dy, dx, lut_i8, interval=4):
    """int16 numerators [C,oC,h,w] (value * 2^interval) of one LUT pass (FourSimplexInterpFaster core)."""
    torch = _torch()
    if img_u8_chw.dtype != torch.uint8 or img_u8_chw.dim() != 3:
        raise ValueError("img must be uint8 [C,H,W]")
    if not 1 <= int(interval) <= 7:
        raise ValueError("interval must be 1..7")
    if lut_i8.dtype != torch.int8 or lut_i8.dim() != 2 or lut_i8.shape[0] != (2 ** (8 - int(interval)) + 1) ** 4:
        raise ValueError("lut must be int8 [L^4,oC] with L = 2^(8-interval) + 1")
    img = img_u8_chw.contiguous()
    lut = lut_i8.contiguous()
    Cn, Hp, Wp = img.shape
    oC = lut.shape[1]
    out = torch.empty((Cn, oC, h, w), dtype=torch.int16, device=img.device)
    dy = np.ascontiguousarray(dy, dtype=np.int8)
    dx = np.ascontiguousarray(dx, dtype=np.int8)
    p = _planes_chw(img)
    _lib.check(_lib.lib().lerf_lut_interp_i16(C.byref(p), Hp, Wp, Cn, int(h), int(w), dy.ctypes.data, dx.ctypes.data,
                                              lut.data_ptr(), oC, int(interval), out.data_ptr(), _lib.current_stream()),
               "lerf_lut_interp_i16")
    return out


# --------------------------------------------------------------------------- A2/A3
def lut_stages(img_u8_hwc, luts):
    """uint8 [H,W,C] -> (feat uint8 [H,W,C], hq uint8 [H,W,C,oC])."""
    torch = _torch()
    if img_u8_hwc.dtype != torch.uint8 or img_u8_hwc.dim() != 3:
        raise ValueError("img must be uint8 [H,W,C]")
    img = img_u8_hwc.contiguous()
    H, W, Cn = img.shape
    feat = torch.empty_like(img)
    hq = torch.empty((H, W, Cn, luts.oC), dtype=torch.uint8, device=img.device)
    pi = _planes_hwc(img)
    pf = _planes_hwc(feat)
    ph = _lib.plane(hq, hq.stride(0), hq.stride(1), hq.stride(2))
    _lib.check(_lib.lib().lerf_lut_stages_u8(C.byref(pi), H, W, Cn, luts.ref(), C.byref(pf), C.byref(ph),
                                             _lib.current_stream()), "lerf_lut_stages_u8")
    return feat, hq


def stages_packed(img_u8, luts, workspace=None):
    """uint8 [H,W,3] / [N,H,W,3] -> int32 [.., H,W,3] packed (hq0 | hq1<<8 | hq2<<16 | feat<<24) by the tile-fused
    stages kernel.  Raises LerfError(unsupported) for configurations it does not cover."""
    torch = _torch()
    if img_u8.dtype != torch.uint8:
        raise ValueError("img must be uint8")
    squeeze = img_u8.dim() == 3
    img = (img_u8.unsqueeze(0) if squeeze else img_u8).contiguous()
    N, H, W, Cn = img.shape
    packed = torch.empty((N, H, W, Cn), dtype=torch.int32, device=img.device)
    need = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N))
    if workspace is None:
        workspace = fused_workspace(H, W, Cn, N, img.device if img.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
    _lib.check(_lib.lib().lerf_stages_packed_u8(img.data_ptr(), img.stride(0), N, H, W, Cn, luts.ref(),
                                                packed.data_ptr(), packed.stride(0), workspace.data_ptr(),
                                                _lib.current_stream()), "lerf_stages_packed_u8")
    return packed[0] if squeeze else packed


def unpack_stages(packed, oC):
    torch = _torch()
    feat = torch.empty(tuple(packed.shape), dtype=torch.uint8, device=packed.device)
    hq = torch.empty(tuple(packed.shape) + (oC,), dtype=torch.uint8, device=packed.device)
    p = packed.contiguous()
    _lib.check(_lib.lib().lerf_unpack_stages(p.data_ptr(), p.numel(), int(oC), feat.data_ptr(), hq.data_ptr(),
                                             _lib.current_stream()), "lerf_unpack_stages")
    return feat, hq


def warp_packed(packed_hwc, geo: "WarpGeometry", kind="gauss", max_sigma=10.0, out="u8"):
    """out: "u8" / "f32" (a fresh tensor) or a caller-owned uint8 / float32 tensor [oH,oW,C] to write into."""
    torch = _torch()
    p = packed_hwc.contiguous()
    H, W, Cn = p.shape
    if isinstance(out, str):
        o = torch.empty((geo.out_hw[0], geo.out_hw[1], Cn), dtype=_out_dtype(out), device=p.device)
    else:
        o = out
        if tuple(o.shape) != (geo.out_hw[0], geo.out_hw[1], Cn) or o.dtype not in (torch.uint8, torch.float32) \
                or o.device != p.device or o.stride(2) != 1:
            raise ValueError("out must be a uint8/float32 [oH,oW,C] tensor on the input's device")
    po = _planes_hwc(o)
    _lib.check(_lib.lib().lerf_warp_packed(p.data_ptr(), H, W, Cn, geo.ref(), KINDS[kind], float(max_sigma),
                                           C.byref(po), _lib.current_stream()), "lerf_warp_packed")
    return o


# --------------------------------------------------------------------------- A5/A6/A8
def _hyper_planes(hyper, layout, nh):
    arr = (_lib.Plane * 3)()
    keep = []
    if layout == "hwck":          # one uint8 tensor [H,W,C,oC]
        hq = hyper
        for k in range(3):
            arr[k] = _lib.plane(hq, hq.stride(0), hq.stride(1), hq.stride(2), offset=(k if k < nh else 0) * hq.stride(3))
        keep.append(hq)
    else:                          # separate planar [N,H,W] tensors with identical strides
        for k in range(3):
            t = hyper[k if k < nh else 0]
            arr[k] = _planes_chw(t)
        keep.extend(hyper)
    return arr, keep


def resize_hwc_u8(feat_u8, hq_u8, geo: SrGeometry, kind="gauss", max_sigma=10.0, out="u8"):
    """stage 3 on the uint8 stage outputs: feat [H,W,C], hq [H,W,C,oC] -> [oH,oW,C]."""
    torch = _torch()
    feat = feat_u8.contiguous()
    H, W, Cn = feat.shape
    nh = {"gauss": 3, "linear": 1}.get(kind, 0)          # fixed kernels (cubic, bilinear, ...) take no hyper maps
    o = torch.empty((geo.out_hw[0], geo.out_hw[1], Cn), dtype=_out_dtype(out), device=feat.device)
    pf = _planes_hwc(feat)
    if nh:
        hq = hq_u8.contiguous()
        if hq.shape[:3] != feat.shape or hq.shape[3] < nh:
            raise ValueError("hyper shape mismatch")
        ph, _keep = _hyper_planes(hq, "hwck", nh)
    else:
        ph = None
    po = _planes_hwc(o)
    _lib.check(_lib.lib().lerf_resize(C.byref(pf), ph, H, W, Cn, geo.ref(), KINDS[kind], float(max_sigma),
                                      C.byref(po), _lib.current_stream()), "lerf_resize")
    return o


def resize_planar(feat, hypers, geo: SrGeometry, kind="gauss", max_sigma=10.0, out="f32"):
    """stage 3 on planar float32 maps: feat [N,H,W], hypers = list of [N,H,W] in [0,1] -> [N,oH,oW]."""
    torch = _torch()
    feat = feat.contiguous().float()
    nh = {"gauss": 3, "linear": 1}.get(kind, 0)
    N, H, W = feat.shape
    o = torch.empty((N, geo.out_hw[0], geo.out_hw[1]), dtype=_out_dtype(out), device=feat.device)
    pf = _planes_chw(feat)
    if nh:
        hypers = [h.contiguous().float() for h in hypers[:nh]]
        for h in hypers:
            if h.shape != feat.shape:
                raise ValueError("hyper maps must have the shape of the input")
        ph, _keep = _hyper_planes(hypers, "planar", nh)
    else:
        ph = None
    po = _planes_chw(o)
    _lib.check(_lib.lib().lerf_resize(C.byref(pf), ph, H, W, N, geo.ref(), KINDS[kind], float(max_sigma),
                                      C.byref(po), _lib.current_stream()), "lerf_resize")
    return o


def warp_hwc_u8(feat_u8, hq_u8, geo: WarpGeometry, kind="gauss", max_sigma=10.0, out="u8"):
    torch = _torch()
    feat = feat_u8.contiguous()
    H, W, Cn = feat.shape
    nh = {"gauss": 3, "linear": 1}.get(kind, 0)
    o = torch.empty((geo.out_hw[0], geo.out_hw[1], Cn), dtype=_out_dtype(out), device=feat.device)
    pf = _planes_hwc(feat)
    if nh:
        hq = hq_u8.contiguous()
        ph, _keep = _hyper_planes(hq, "hwck", nh)
    else:
        ph = None
    po = _planes_hwc(o)
    _lib.check(_lib.lib().lerf_warp(C.byref(pf), ph, H, W, Cn, geo.ref(), KINDS[kind], float(max_sigma),
                                    C.byref(po), _lib.current_stream()), "lerf_warp")
    return o


def warp_planar(feat, hypers, geo: WarpGeometry, kind="gauss", max_sigma=10.0, out="f32"):
    torch = _torch()
    feat = feat.contiguous().float()
    nh = {"gauss": 3, "linear": 1}.get(kind, 0)
    N, H, W = feat.shape
    o = torch.empty((N, geo.out_hw[0], geo.out_hw[1]), dtype=_out_dtype(out), device=feat.device)
    pf = _planes_chw(feat)
    if nh:
        hypers = [h.contiguous().float() for h in hypers[:nh]]
        ph, _keep = _hyper_planes(hypers, "planar", nh)
    else:
        ph = None
    po = _planes_chw(o)
    _lib.check(_lib.lib().lerf_warp(C.byref(pf), ph, H, W, N, geo.ref(), KINDS[kind], float(max_sigma),
                                    C.byref(po), _lib.current_stream()), "lerf_warp")
    return o


# --------------------------------------------------------------------------- fused SR
_WS = {}


def fused_workspace(H, W, Cn, N, device):
    """Device scratch of lerf_sr_fused_workspace_bytes() bytes for the two-launch fused path (stage-1 output of the
    batch between s1_kernel and the stage-2/3 launch), cached per device and grown on demand: repeated calls on one
    stream reuse it (launches on a stream are ordered, so the previous call has consumed it).  Callers that run
    several streams concurrently pass their own `workspace=`."""
    torch = _torch()
    need = max(int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N)), 1)
    key = (device.type, device.index)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _gpu_visible(t):
    """device memory, or pinned host memory (the kernels read / write it over PCIe: stream.StreamingSR)"""
    return t.is_cuda or t.is_pinned()


def _check_out_u8(out, shape, device, what):
    torch = _torch()
    if not isinstance(out, torch.Tensor) or out.dtype != torch.uint8 or tuple(out.shape) != tuple(shape):
        raise ValueError("%s must be a uint8 tensor of shape %s" % (what, tuple(shape)))
    if not _gpu_visible(out):
        raise ValueError("%s must live in device memory or pinned host memory" % what)
    exp = 1
    for d in range(out.dim() - 1, 0, -1):               # frames may be strided (dim 0); each frame is dense HWC
        if out.shape[d] != 1 and out.stride(d) != exp:
            raise ValueError("%s: every frame must be contiguous [H,W,C]" % what)
        exp *= out.shape[d]


def sr_fused_u8(img_u8, luts, geo: SrGeometry, kind="gauss", max_sigma=10.0, out=None, workspace=None):
    """uint8 [H,W,C] or [N,H,W,C] -> uint8 [oH,oW,C] / [N,oH,oW,C].  Two launches per call (stage 1 over the batch into
    the workspace, then stages 2+3 per tile); `out` (same rank as the input) and `workspace` may be caller-owned."""
    torch = _torch()
    if img_u8.dtype != torch.uint8:
        raise ValueError("img must be uint8")
    squeeze = img_u8.dim() == 3
    img = (img_u8.unsqueeze(0) if squeeze else img_u8)
    if img.dim() != 4:
        raise ValueError("img must be [H,W,C] or [N,H,W,C]")
    if not img[0].is_contiguous():
        img = img.contiguous()
    N, H, W, Cn = img.shape
    if (H, W) != geo.in_hw:
        raise ValueError("geometry was built for another input size")
    oshape = (N, geo.out_hw[0], geo.out_hw[1], Cn)
    if not _gpu_visible(img):
        raise ValueError("img must live in device memory or pinned host memory")
    if out is None:
        o4 = torch.empty(oshape, dtype=torch.uint8, device=img.device if img.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    else:
        o4 = out.unsqueeze(0) if (squeeze and out.dim() == 3) else out
        _check_out_u8(o4, oshape, img.device, "out")
    need = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N))
    if workspace is None:
        workspace = fused_workspace(H, W, Cn, N, img.device if img.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
    _lib.check(_lib.lib().lerf_sr_fused_u8(img.data_ptr(), img.stride(0), N, H, W, Cn, luts.ref(), geo.ref(),
                                           KINDS[kind], float(max_sigma), o4.data_ptr(), o4.stride(0),
                                           workspace.data_ptr(), _lib.current_stream()), "lerf_sr_fused_u8")
    return o4[0] if squeeze else o4
