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
        # exact x2 tables of lerf_sr_axis_tables on both axes: rows / columns pair up on their taps, distances have period 2 --
        # lets lerf_sr_fused_u8 take the persistent kernel (LERF_GEO_X2_TABLES; slices of these tables keep the property)
        if sh == 2.0 and sw == 2.0 and arithmetic == "f64" and float(dis_scale) == 1.0 and self.S == 2 \
                and self.out_hw == (2 * H, 2 * W):
            self.flags = _lib.GEO_X2_TABLES
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
        g.tie_queue_cap = int(getattr(self, "tie_queue_cap", 0))
        g.roi_y, g.roi_x, g.roi_h, g.roi_w = getattr(self, "roi", (0, 0, 0, 0))
        g.flags = int(getattr(self, "flags", 0))
        for k in ("left_r", "dis_r", "left_c", "dis_c", "dis_r64", "dis_c64"):
            setattr(g, k, self.t[k].data_ptr())
        self.struct = g

    def with_tie_queue_cap(self, cap):
        """Test hook (lerf_sr_geo_t.tie_queue_cap): a copy of this geometry whose launches queue at most `cap` rounding
        ties per tile (0 entries: cap < 0 in the ABI; None restores the default) -- drives the in-loop fallback."""
        g = object.__new__(SrGeometry)
        g.__dict__.update(self.__dict__)
        g.tie_queue_cap = 0 if cap is None else (-1 if int(cap) == 0 else int(cap))
        g._upload()
        return g

    def with_flags(self, flags):
        """a copy of this geometry with `flags` added to lerf_sr_geo_t.flags (_lib.GEO_*: diagnostic A/B of the kernel families)"""
        g = object.__new__(SrGeometry)
        g.__dict__.update(self.__dict__)
        g.flags = int(getattr(self, "flags", 0)) | int(flags)
        g._upload()
        return g

    def _check_partition_pad(self):
        # a slice pads at the LOCAL frame borders: wrap padding would wrap inside the strip / block, and the far side of
        # the frame lives on another rank.  constant / edge / reflect / symmetric only reach pixels next to the global border,
        # which a slice that touches that border holds itself.
        if self.pad_mode == _lib.PAD_MODES["wrap"]:
            raise ValueError("strip / block partitions do not support pad_mode 'wrap' (the far side of the frame is on another rank)")

    def row_slice(self, lr_row0, lr_rows, out_row0, out_row1):
        """Geometry of a horizontal strip: the LR rows [lr_row0, lr_row0 + lr_rows) held locally
        (owned rows plus halo) produce the global output rows [out_row0, out_row1).  The row tables
        are the global ones, rebased to the strip -- the kernels never see the strip as a frame of
        its own, so non-integer scales partition exactly like integer ones."""
        self._check_partition_pad()
        g = object.__new__(SrGeometry)
        g.in_hw = (int(lr_rows), self.in_hw[1])
        g.out_hw = (int(out_row1 - out_row0), self.out_hw[1])
        g.scales, g.S, g.device, g.pad_vec = self.scales, self.S, self.device, self.pad_vec
        g.pad_mode = self.pad_mode
        g.flags = int(getattr(self, "flags", 0))
        h = self.host
        g.host = dict(left_r=(h["left_r"][out_row0:out_row1] - lr_row0).astype(np.int32),
                      dis_r=h["dis_r"][out_row0:out_row1], dis_r32=h["dis_r32"][out_row0:out_row1],
                      left_c=h["left_c"], dis_c=h["dis_c"], dis_c32=h["dis_c32"])
        g._upload()
        return g

    def block_slice(self, lr_row0, lr_rows, out_row0, out_row1, lr_col0, lr_cols, out_col0, out_col1, roi=None):
        """Geometry of a 2-D block: the LR rows [lr_row0, +lr_rows) x columns [lr_col0, +lr_cols) held locally (owned block
        plus halo) produce the global output rows [out_row0, out_row1) x columns [out_col0, out_col1).  Both axes' tables
        are the global ones rebased to the block.  roi = (y, x, h, w) of the OWNED block inside the local frame: the
        tile-fused kernel lays its tiles over it, so halo pixels only ever serve as tile halos."""
        g = self.row_slice(lr_row0, lr_rows, out_row0, out_row1)
        h = self.host
        g.in_hw = (int(lr_rows), int(lr_cols))
        g.out_hw = (int(out_row1 - out_row0), int(out_col1 - out_col0))
        g.host.update(left_c=(h["left_c"][out_col0:out_col1] - lr_col0).astype(np.int32),
                      dis_c=h["dis_c"][out_col0:out_col1], dis_c32=h["dis_c32"][out_col0:out_col1])
        if roi is not None:
            g.roi = tuple(int(v) for v in roi)
        g._upload()
        return g

    def ref(self):
        return C.byref(self.struct)


class WarpGeometry:
    """Homography geometry (Warp2dNumpy.set_shape, resize_right2d_numpy.py:292-407)."""

    def __init__(self, in_hw, matrix, out_hw, support=2, pad_mode=0, out_rect=None, src_y0=0):
        """out_rect = (i0, i1, j0, j1): the geometry of that rectangle of the output alone (`out_hw` stays the WHOLE output's: the
        pads come from its corner pixels); src_y0: the source operands hold the frame's rows from src_y0 on (a rank's band of a
        partitioned warp, dist.WarpRowPlan).  Same float64 arithmetic as the whole frame's, bit for bit."""
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
        self.full_out_hw = self.out_hw
        if out_rect is not None:
            i0, i1, j0, j1 = (int(v) for v in out_rect)
            if not (0 <= i0 < i1 <= self.out_hw[0] and 0 <= j0 < j1 <= self.out_hw[1]):
                raise ValueError("out_rect outside the output")
            g.out_y0, g.out_x0, g.out_h, g.out_w = i0, j0, i1 - i0, j1 - j0
            self.out_hw = (i1 - i0, j1 - j0)
        g.src_y0 = self.src_y0 = int(src_y0)
        self.struct = g

    def ref(self):
        return C.byref(self.struct)

    def tile_boxes(self, device):
        """int32 [tiles][4] on `device`: the output rows x columns that bound what each 64 x 64 source tile owns in the
        tile-fused warp (lerf_warp_tile_boxes: one host pass over the output per homography, kept with the geometry)"""
        torch = _torch()
        key = str(device)
        cache = self.__dict__.setdefault("_boxes", {})
        if key not in cache:
            H, W = self.in_hw
            nt = ((H + 63) // 64) * ((W + 63) // 64)
            b = np.zeros((nt, 4), dtype=np.int32)
            _lib.check(_lib.lib().lerf_warp_tile_boxes(self.ref(), H, W, b.ctypes.data), "lerf_warp_tile_boxes")
            cache[key] = (torch.from_numpy(b).to(device), b)
        return cache[key][0]


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


INTERP_ACCUMULATE, INTERP_LDS, INTERP_DIRECT, INTERP_TILE64, INTERP_TILE32, INTERP_LUT_PLANAR = 1, 2, 4, 8, 16, 32     # LERF_INTERP_*
LUT_PLANE_BYTES = 83584                                                                             # LERF_LUT_PLANE_BYTES


def lut_planes(lut_i8):
    """[17^4, oC] int8 -> the LDS kernel's own layout, [oC, 83584] (LERF_INTERP_LUT_PLANAR): a workgroup then copies its plane
    alone instead of reading all oC interleaved bytes of every entry"""
    torch = _torch()
    n, oC = lut_i8.shape
    planes = torch.zeros((oC, LUT_PLANE_BYTES), dtype=torch.int8, device=lut_i8.device)
    planes[:, :n] = lut_i8.t()
    return planes



def lut_interp(img_chw, h, w, dy, dx, lut_i8, interval=4, rot=0, out_dtype=None, out=None, accumulate=False, kernel=None, planes=None):
    """One LUT pass with the reference's epilogue in the store (lerf_lut_interp_ex, ABI 7): img uint8 or float32 [C,Hp,Wp] (any
    strides) -> [C*oC, h', w'] = np.rot90(values, rot, [1, 2]) as float64 (default) / float32 VALUES (numerator / 2^interval),
    or the int16 numerators.  The rotation costs nothing: the kernel stores through the strides of the rotated view.
    out: a contiguous [C*oC, h', w'] tensor to write into; accumulate=True: out += result (the call sites' `pred += ...`).
    kernel: None (the library chooses), "lds", "lds64", "lds32" (the LDS kernel, its tile forced) or "direct" (tests, A/B runs).
    planes: lut_planes(lut_i8), handed to the LDS kernel when it takes the call (interval 4)."""
    torch = _torch()
    if img_chw.dtype not in (torch.uint8, torch.float32) or img_chw.dim() != 3:
        raise ValueError("img must be uint8 or float32 [C,H,W]")
    if not 1 <= int(interval) <= 7:
        raise ValueError("interval must be 1..7")
    if lut_i8.dtype != torch.int8 or lut_i8.dim() != 2 or lut_i8.shape[0] != (2 ** (8 - int(interval)) + 1) ** 4:
        raise ValueError("lut must be int8 [L^4,oC] with L = 2^(8-interval) + 1")
    out_dtype = out_dtype or (out.dtype if out is not None else torch.float64)
    if out_dtype not in (torch.float64, torch.float32, torch.int16):
        raise ValueError("out_dtype must be float64, float32 or int16")
    lut = lut_i8.contiguous()
    Cn, Hp, Wp = img_chw.shape
    oC = lut.shape[1]
    h, w, rot = int(h), int(w), int(rot) % 4
    oh, ow = (h, w) if rot % 2 == 0 else (w, h)
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs the tensor to add into (out=)")
        out = torch.empty((Cn * oC, oh, ow), dtype=out_dtype, device=img_chw.device)
    elif tuple(out.shape) != (Cn * oC, oh, ow) or out.dtype != out_dtype or not out.is_contiguous() or out.device != img_chw.device:
        raise ValueError("out must be a contiguous %s tensor of shape %r on the image's device" % (out_dtype, (Cn * oC, oh, ow)))
    flags = (INTERP_ACCUMULATE if accumulate else 0) | {None: 0, "lds": INTERP_LDS, "lds64": INTERP_LDS | INTERP_TILE64, "lds32": INTERP_LDS | INTERP_TILE32,
                                                      "direct": INTERP_DIRECT}[kernel]
    # element (y, x) of the un-rotated result lands at R = rot90(A, rot): rot 1: R[w-1-x, y]; 2: R[h-1-y, w-1-x]; 3: R[x, h-1-y]
    sy, sx, off = {0: (ow, 1, 0), 1: (1, -ow, (w - 1) * ow), 2: (-ow, -1, h * w - 1), 3: (-1, ow, ow - 1)}[rot]
    po = _lib.plane(out, sy, sx, oh * ow, offset=off)
    dy = np.ascontiguousarray(dy, dtype=np.int8)
    dx = np.ascontiguousarray(dx, dtype=np.int8)
    p = _planes_chw(img_chw)
    with _lib.on_device(out):
        if planes is not None and int(interval) == 4 and kernel != "direct" and oC > 1:
            if planes.dtype != torch.int8 or tuple(planes.shape) != (oC, LUT_PLANE_BYTES) or not planes.is_contiguous():
                raise ValueError("planes must be lut_planes(lut_i8)")
            rc = _lib.lib().lerf_lut_interp_ex(C.byref(p), Hp, Wp, Cn, h, w, dy.ctypes.data, dx.ctypes.data, planes.data_ptr(), oC,
                                               4, C.byref(po), flags | INTERP_LUT_PLANAR, _lib.current_stream())
            if rc != -2 or kernel is not None:             # LERF_EUNSUPPORTED: the direct kernel serves the call from the interleaved table
                _lib.check(rc, "lerf_lut_interp_ex")
                return out
        _lib.check(_lib.lib().lerf_lut_interp_ex(C.byref(p), Hp, Wp, Cn, h, w, dy.ctypes.data, dx.ctypes.data, lut.data_ptr(), oC,
                                                 int(interval), C.byref(po), flags, _lib.current_stream()), "lerf_lut_interp_ex")
    return out


EPI_DIV, EPI_MUL, EPI_ADD, EPI_CLIP, EPI_ROUND = 0, 1, 2, 3, 4     # LERF_EPI_* of include/lerf_hip.h


def numer_epilogue(acc_i16, interval, steps):
    """int16 numerators (value * 2^interval) -> float32, through a program of float64 steps [(EPI_*, a, b), ...] in numpy's
    order (lerf_numer_epilogue_f32): np.round(np.clip(pred / n + bias, 0, norm)).astype(np.float32) of the call sites"""
    torch = _torch()
    if acc_i16.dtype != torch.int16 or not acc_i16.is_contiguous() or len(steps) > 8:
        raise ValueError("acc must be a contiguous int16 tensor, at most 8 steps")
    out = torch.empty(acc_i16.shape, dtype=torch.float32, device=acc_i16.device)
    prog = (_lib.EpiOp * max(len(steps), 1))()
    for k, st in enumerate(steps):
        prog[k].op, prog[k].a, prog[k].b = int(st[0]), float(st[1]) if len(st) > 1 else 0.0, float(st[2]) if len(st) > 2 else 0.0
    with _lib.on_device(out):
        _lib.check(_lib.lib().lerf_numer_epilogue_f32(acc_i16.data_ptr(), acc_i16.numel(), int(interval), C.addressof(prog), len(steps),
                                                      out.data_ptr(), _lib.current_stream()), "lerf_numer_epilogue_f32")
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
    with _lib.on_device(packed):
        _lib.check(_lib.lib().lerf_stages_packed_u8(img.data_ptr(), img.stride(0), N, H, W, Cn, luts.ref(),
                                                    packed.data_ptr(), packed.stride(0), workspace.data_ptr(), workspace.numel(),
                                                    _lib.current_stream()), "lerf_stages_packed_u8")
    return packed[0] if squeeze else packed


def stages_packed_ragged(imgs_u8, luts, workspace=None):
    """Frames of DIFFERENT sizes, one launch pair (lerf_stages_packed_ragged_u8): list of uint8 [H_i,W_i,C] device tensors
    -> list of int32 [H_i,W_i,C] packed stage outputs."""
    torch = _torch()
    if not imgs_u8:
        return []
    xs = [x.contiguous() for x in imgs_u8]
    Cn = xs[0].shape[-1]
    for x in xs:
        if x.dtype != torch.uint8 or x.dim() != 3 or x.shape[-1] != Cn or not x.is_cuda or x.device != xs[0].device:
            raise ValueError("ragged stages take uint8 [H,W,C] tensors of ONE device with one channel count")
    outs = [torch.empty(tuple(x.shape), dtype=torch.int32, device=x.device) for x in xs]
    items = (_lib.StageItem * len(xs))()
    for it, x, o in zip(items, xs, outs):
        it.img, it.packed, it.H, it.W = x.data_ptr(), o.data_ptr(), x.shape[0], x.shape[1]
    need = int(_lib.lib().lerf_stages_ragged_workspace_bytes(items, len(xs), Cn))
    if workspace is None:
        workspace = _cached_workspace(need, xs[0].device)
    elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
    with _lib.on_device(xs[0]):
        _lib.check(_lib.lib().lerf_stages_packed_ragged_u8(items, len(xs), Cn, luts.ref(), workspace.data_ptr(), workspace.numel(),
                                                           _lib.current_stream()), "lerf_stages_packed_ragged_u8")
    return outs


def unpack_stages(packed, oC):
    torch = _torch()
    feat = torch.empty(tuple(packed.shape), dtype=torch.uint8, device=packed.device)
    hq = torch.empty(tuple(packed.shape) + (oC,), dtype=torch.uint8, device=packed.device)
    p = packed.contiguous()
    _lib.check(_lib.lib().lerf_unpack_stages(p.data_ptr(), p.numel(), int(oC), feat.data_ptr(), hq.data_ptr(),
                                             _lib.current_stream()), "lerf_unpack_stages")
    return feat, hq


def warp_packed(packed, geo: "WarpGeometry", kind="gauss", max_sigma=10.0, out="u8"):
    """packed: int32 [H,W,C] or a batch [N,H,W,C] sharing the homography (ONE launch for the batch).
    out: "u8" / "f32" (a fresh tensor) or a caller-owned uint8 / float32 tensor of the output shape to write into."""
    torch = _torch()
    squeeze = packed.dim() == 3
    p = packed.unsqueeze(0) if squeeze else packed
    if not p[0].is_contiguous():
        p = p.contiguous()
    N, Hb, W, Cn = p.shape
    H = geo.in_hw[0]                                        # the frame's height; `packed` may hold its rows from geo.src_y0 on only
    if W != geo.in_hw[1] or geo.src_y0 + Hb > H:
        raise ValueError("packed maps do not match the geometry's frame")
    oshape = (N, geo.out_hw[0], geo.out_hw[1], Cn)
    if isinstance(out, str):
        o = torch.empty(oshape, dtype=_out_dtype(out), device=p.device)
    else:
        o = out.unsqueeze(0) if (squeeze and out.dim() == 3) else out
        if tuple(o.shape) != oshape or o.dtype not in (torch.uint8, torch.float32) or o.device != p.device \
                or o.stride(3) != 1:
            raise ValueError("out must be a uint8/float32 tensor of shape %s on the input's device" % (oshape[1:] if squeeze else oshape,))
    po = _lib.plane(o, o.stride(1), o.stride(2), o.stride(3))
    _lib.check(_lib.lib().lerf_warp_packed(p.data_ptr(), p.stride(0), N, H, W, Cn, geo.ref(), KINDS[kind], float(max_sigma),
                                           C.byref(po), o.stride(0), _lib.current_stream()), "lerf_warp_packed")
    return o[0] if squeeze else o


def warp_fused_supported(img_u8, luts, geo: "WarpGeometry", kind="gauss", max_sigma=10.0):
    H, W, Cn = img_u8.shape[-3:]
    return bool(_lib.lib().lerf_warp_fused_supported(Cn, luts.ref(), geo.ref(), H, W, KINDS[kind], float(max_sigma)))


def warp_fused_u8(img_u8, luts, geo: "WarpGeometry", kind="gauss", max_sigma=10.0, out=None, workspace=None):
    """The whole warp path tile-fused (lerf_warp_fused_u8): uint8 [H,W,3] or a batch [N,H,W,3] sharing the homography ->
    uint8 [.., oH, oW, 3]; stage 1 into the workspace, then stage 2 + the warp per source tile -- no packed maps in HBM."""
    torch = _torch()
    if img_u8.dtype != torch.uint8:
        raise ValueError("img must be uint8")
    squeeze = img_u8.dim() == 3
    img = (img_u8.unsqueeze(0) if squeeze else img_u8).contiguous()
    N, H, W, Cn = img.shape
    oshape = (N, geo.out_hw[0], geo.out_hw[1], Cn)
    if out is None:
        o = torch.empty(oshape, dtype=torch.uint8, device=img.device)
    else:
        o = out.unsqueeze(0) if (squeeze and out.dim() == 3) else out
        if tuple(o.shape) != oshape or o.dtype != torch.uint8 or not o.is_contiguous() or o.device != img.device:
            raise ValueError("out must be a contiguous uint8 tensor of shape %s on the input's device" % (oshape,))
    need = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N))
    if workspace is None:
        workspace = fused_workspace(H, W, Cn, N, img.device)
    elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
    boxes = geo.tile_boxes(img.device)
    with _lib.on_device(o):
        _lib.check(_lib.lib().lerf_warp_fused_u8(img.data_ptr(), img.stride(0), N, H, W, Cn, luts.ref(), geo.ref(), boxes.data_ptr(),
                                                 KINDS[kind], float(max_sigma), o.data_ptr(), o.stride(0), workspace.data_ptr(),
                                                 workspace.numel(), _lib.current_stream()), "lerf_warp_fused_u8")
    return o[0] if squeeze else o


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


def resize_planar_u8(feat_u8, hq_u8, geo: SrGeometry, kind="gauss", max_sigma=10.0):
    """stage 3 on planar uint8 maps: feat [N,H,W] (any strides), hq = list of uint8 numerator maps [N,H,W] with identical
    strides (hyper = hq / 255) -> uint8 [oH,oW,N] = clip(rne(value), 0, 255): the production arithmetic of the fused path"""
    torch = _torch()
    nh = {"gauss": 3, "linear": 1}[kind]
    N, H, W = feat_u8.shape
    hq = [h.contiguous() for h in hq_u8[:nh]]
    if feat_u8.dtype != torch.uint8 or any(h.dtype != torch.uint8 or h.shape != feat_u8.shape for h in hq):
        raise ValueError("uint8 maps of the input's shape")
    o = torch.empty((geo.out_hw[0], geo.out_hw[1], N), dtype=torch.uint8, device=feat_u8.device)
    pf = _planes_chw(feat_u8)
    ph, _keep = _hyper_planes(hq, "planar", nh)
    po = _planes_hwc(o)
    with _lib.on_device(o):
        _lib.check(_lib.lib().lerf_resize(C.byref(pf), ph, H, W, N, geo.ref(), KINDS[kind], float(max_sigma),
                                          C.byref(po), _lib.current_stream()), "lerf_resize")
    return o


def warp_hwc_u8(feat_u8, hq_u8, geo: WarpGeometry, kind="gauss", max_sigma=10.0, out="u8"):
    torch = _torch()
    feat = feat_u8.contiguous()
    Hb, W, Cn = feat.shape
    H = geo.in_hw[0]                                        # (a band of the frame from geo.src_y0 on: see WarpGeometry)
    if W != geo.in_hw[1] or geo.src_y0 + Hb > H:
        raise ValueError("the maps do not match the geometry's frame")
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
_WS_MAX = 8


def _cached_workspace(need, device):
    """Scratch of at least `need` bytes, cached per (device, stream): launches on ONE stream are ordered, so the previous call
    on that stream has consumed the buffer; another stream (or thread with its own stream) gets a buffer of its own instead
    of racing for this one."""
    torch = _torch()
    need = max(int(need), 1)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    # the stream the launch will use: every launch below runs inside `_lib.on_device(tensor)` and passes
    # _lib.current_stream() of that device -- the same call that forms this key
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    ws = _WS.pop(key, None)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
    _WS[key] = ws                      # most recently used last
    while len(_WS) > _WS_MAX:          # streams come and go (stream handles are recycled): keep the cache bounded
        _WS.pop(next(iter(_WS)))
    return ws


def fused_workspace(H, W, Cn, N, device):
    """Device scratch of lerf_sr_fused_workspace_bytes() bytes for the two-launch fused path (stage-1 output of the
    batch between s1_kernel and the stage-2/3 launch), cached per device AND stream and grown on demand."""
    return _cached_workspace(int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N)), device)


def _gpu_visible(t):
    """device memory, or pinned host memory (the kernels read / write it over PCIe: stream.StreamingSR)"""
    return t.is_cuda or t.is_pinned()


def _check_out_u8(out, shape, device, what, pitched=False):
    """frames may be strided (dim 0); each frame is dense HWC -- or, with pitched=True, HWC rows at a pitch >= W * C bytes
    (a view [:, :, :W] of a wider tensor: lerf_sr_geo_t.out_row_pitch)"""
    torch = _torch()
    if not isinstance(out, torch.Tensor) or out.dtype != torch.uint8 or tuple(out.shape) != tuple(shape):
        raise ValueError("%s must be a uint8 tensor of shape %s" % (what, tuple(shape)))
    if not _gpu_visible(out):
        raise ValueError("%s must live in device memory or pinned host memory" % what)
    exp = 1
    for d in range(out.dim() - 1, 0, -1):
        if out.shape[d] != 1 and out.stride(d) != exp:
            if pitched and d == out.dim() - 3 and out.stride(d) > exp:
                exp = out.stride(d)                     # the row pitch
            else:
                raise ValueError("%s: every frame must be contiguous [H,W,C]%s" % (what, " (rows may be pitched)" if pitched else ""))
        exp *= out.shape[d]


def sr_fused_u8(img_u8, luts, geo: SrGeometry, kind="gauss", max_sigma=10.0, out=None, workspace=None):
    """uint8 [H,W,C] or [N,H,W,C] -> uint8 [oH,oW,C] / [N,oH,oW,C].  Two launches per call (stage 1 over the batch into
    the workspace, then stages 2+3 per tile); `out` (same rank as the input) and `workspace` may be caller-owned.
    workspace=False: ONE launch, every tile recomputes stage 1 on its halo -- the better choice for a single frame or
    block that fills the chip once (no workspace, no second launch)."""
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
    dev = img.device if img.is_cuda else torch.device("cuda", torch.cuda.current_device())
    if out is None:
        o4 = torch.empty(oshape, dtype=torch.uint8, device=dev)
    else:
        o4 = out.unsqueeze(0) if (squeeze and out.dim() == 3) else out
        _check_out_u8(o4, oshape, img.device, "out", pitched=True)
    need = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, Cn, N))
    if workspace is False:
        ws_ptr, ws_n = None, 0
    else:
        if workspace is None:
            workspace = fused_workspace(H, W, Cn, N, dev)
        elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
            raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
        ws_ptr, ws_n = workspace.data_ptr(), workspace.numel()
    # where the frames live travels with the call (no hipPointerGetAttributes per launch)
    gs = _lib.SrGeo.from_buffer_copy(geo.struct)
    gs.flags |= _lib.GEO_INPUT_DEVICE if img.is_cuda else _lib.GEO_INPUT_HOST
    if o4.shape[1] > 1 and o4.stride(1) != oshape[2] * Cn:
        gs.out_row_pitch = int(o4.stride(1))         # rows of a wider tensor (dist.sr_block pads a block's rows to 16 bytes)
    with _lib.on_device(o4):
        _lib.check(_lib.lib().lerf_sr_fused_u8(img.data_ptr(), img.stride(0), N, H, W, Cn, luts.ref(), C.byref(gs),
                                               KINDS[kind], float(max_sigma), o4.data_ptr(), o4.stride(0),
                                               ws_ptr, ws_n, _lib.current_stream()), "lerf_sr_fused_u8")
    return o4[0] if squeeze else o4


def sr_fused_supported(Cn, luts, geo: SrGeometry, kind="gauss", max_sigma=10.0):
    """True when lerf_sr_fused_u8 takes the tile-fused kernels for this configuration (else: the three direct kernels)."""
    return bool(_lib.lib().lerf_sr_fused_supported(int(Cn), luts.ref(), geo.ref(), geo.in_hw[0], geo.in_hw[1], KINDS[kind],
                                                   float(max_sigma)))


def sr_fused_ragged_u8(imgs_u8, luts, geos, kind="gauss", max_sigma=10.0, workspace=None):
    """Frames of DIFFERENT sizes through one launch pair (lerf_sr_fused_ragged_u8): lists of uint8 [H_i,W_i,C] device
    tensors and their SrGeometry -> list of uint8 [oH_i,oW_i,C].  What the reference's harness does image by image over a
    benchmark folder (eval_lut_sr.py:489-512)."""
    torch = _torch()
    if len(imgs_u8) != len(geos) or not imgs_u8:
        raise ValueError("one geometry per image")
    xs = [x.contiguous() for x in imgs_u8]
    Cn = xs[0].shape[-1]
    for x, g in zip(xs, geos):
        if x.dtype != torch.uint8 or x.dim() != 3 or x.shape[-1] != Cn or not x.is_cuda or x.device != xs[0].device:
            raise ValueError("ragged SR takes uint8 [H,W,C] tensors of ONE device with one channel count")
        if g.device != xs[0].device and (g.device.index is not None or xs[0].device.index != torch.cuda.current_device()):
            raise ValueError("the geometry tables live on another device than the frames")
        if tuple(x.shape[:2]) != g.in_hw:
            raise ValueError("geometry was built for another input size")
    outs = [torch.empty((g.out_hw[0], g.out_hw[1], Cn), dtype=torch.uint8, device=x.device) for x, g in zip(xs, geos)]
    items = (_lib.SrItem * len(xs))()
    for it, x, o, g in zip(items, xs, outs, geos):
        it.img, it.out, it.H, it.W, it.geo = x.data_ptr(), o.data_ptr(), x.shape[0], x.shape[1], g.struct
    need = int(_lib.lib().lerf_sr_ragged_workspace_bytes(items, len(xs), Cn))
    if workspace is None:
        workspace = _cached_workspace(need, xs[0].device)
    elif workspace.dtype != torch.uint8 or workspace.numel() < need or not workspace.is_cuda or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous uint8 device tensor of at least %d bytes" % need)
    with _lib.on_device(xs[0]):
        _lib.check(_lib.lib().lerf_sr_fused_ragged_u8(items, len(xs), Cn, luts.ref(), KINDS[kind], float(max_sigma),
                                                      workspace.data_ptr(), workspace.numel(), _lib.current_stream()),
                   "lerf_sr_fused_ragged_u8")
    return outs


def rect_copy(frames_u8, staging_u8, rects, to_staging):
    """lerf_rect_copy_u8: rectangles (y, x, h, w, byte offset in staging) of a dense uint8 batch [N,fh,fw,C] <-> one
    contiguous staging tensor, ONE launch (halo pack / unpack of dist.BlockBuffer)."""
    torch = _torch()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or not frames_u8.is_contiguous() or not staging_u8.is_contiguous():
        raise ValueError("frames must be a dense uint8 [N,fh,fw,C] tensor, staging contiguous")
    if not 1 <= len(rects) <= _lib.LERF_MAX_RECTS:
        raise ValueError("1..%d rectangles" % _lib.LERF_MAX_RECTS)
    N, fh, fw, Cn = frames_u8.shape
    arr = (_lib.Rect * len(rects))()
    for a, (y, x, h, w, off) in zip(arr, rects):
        a.y, a.x, a.h, a.w, a.off = int(y), int(x), int(h), int(w), int(off)
        if off + N * h * w * Cn > staging_u8.numel():
            raise ValueError("staging buffer too small")
    with _lib.on_device(frames_u8):
        _lib.check(_lib.lib().lerf_rect_copy_u8(frames_u8.data_ptr(), N, fh, fw, Cn, staging_u8.data_ptr(), arr, len(rects),
                                                1 if to_staging else 0, _lib.current_stream()), "lerf_rect_copy_u8")
