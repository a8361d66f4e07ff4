"""CPU oracle for the LeRF LUT resampling hot path  --  TEST INFRASTRUCTURE ONLY.

This is a from-scratch numpy restatement of the reference algorithm.  It is the
*checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``lerf-pytorch_amd``) never imports anything from ``oracle/``.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against vectors produced by importing the reference itself
(``tests/golden/gen_golden.py``, run in the build container) and against the
Set5 known-answer table of the reference's ``scripts.sh``.

Reference citations are relative to the upstream repository root:

* stage 1/2 LUT interpolation ........ resample/eval_lut_sr.py:24-470  (FourSimplexInterpFaster)
* stage 1/2 rotation ensembles ....... resample/eval_lut_sr.py:541-628, resample/eval_lut_warp.py:104-191
* SR geometry ........................ resize_right/resize_right2d_numpy.py:57-140
* steering-Gaussian resize ........... resize_right/resize_right2d_numpy.py:142-223
* amplified-linear resize ............ resize_right/resize_right2d_numpy.py:225-282
* homography geometry ................ resize_right/resize_right2d_numpy.py:306-407
* warps (gauss / linear / nearest) ... resize_right/resize_right2d_numpy.py:409-449, 460-467, 496-636
* box kernel ......................... resize_right/interp_methods.py:67-70
* metrics ............................ common/utils.py:46-76, 138-151, 168-175
"""
from __future__ import annotations

import math

import numpy as np

Q = 16          # 2**interval, interval = 4  (eval_lut_sr.py:27)
L = 17          # 2**(8-interval) + 1       (eval_lut_sr.py:28)
STRIDES = (L * L * L, L * L, L, 1)
EPS32 = float(np.finfo(np.float32).eps)

# sampling patterns: (dy, dx) of pixels a, b, c, d in the *rotated* frame
# (eval_lut_sr.py:30-81)
MODE_OFFSETS = {
    "s": ((0, 0), (0, 1), (1, 0), (1, 1)),
    "d": ((0, 0), (0, 2), (2, 0), (2, 2)),
    "y": ((0, 0), (1, 1), (1, 2), (2, 1)),
    "c": ((0, 0), (0, 1), (0, 2), (0, 3)),
    "t": ((0, 0), (1, 1), (2, 2), (3, 3)),
}
MODE_PAD = {"s": 1, "d": 2, "y": 2, "c": 3, "t": 3}   # eval_lut_sr.py:12-18


def rotated_offsets(mode: str, r: int):
    """Offsets of the 4 sampled pixels in the UNROTATED frame for rotation r.

    The reference rotates the image by ``np.rot90(img, r)``, pads bottom/right
    with edge replication, samples the pattern, and rotates the result back
    (eval_lut_sr.py:549-553, 468).  In the unrotated frame that is: apply
    (dy, dx) -> (dx, -dy) r times and clamp the coordinates to the image.
    """
    if mode not in MODE_OFFSETS:
        raise ValueError("Mode {} not implemented.".format(mode))
    offs = []
    for dy, dx in MODE_OFFSETS[mode]:
        for _ in range(r % 4):
            dy, dx = dx, -dy
        offs.append((dy, dx))
    return offs


def simplex_numer(lut: np.ndarray, v: np.ndarray, interval: int = 4) -> np.ndarray:
    """4-simplex interpolation numerators (value * 2^interval), exact integers; interval = 4 for the shipped LUTs
    (q = 2^interval, L = 2^(8-interval) + 1 levels per axis, eval_lut_sr.py:27-28).

    lut: int [L**4, oC]; v: int [4, ...] pixel values of a, b, c, d (0..255).
    Returns int32 [..., oC].  Equivalent to the 24 ordered cases of
    eval_lut_sr.py:218-462: walk from the base corner along the axes in order
    of decreasing LSB.  Ties contribute zero weight, so their order is free.
    """
    lut = np.asarray(lut).astype(np.int32)
    v = np.asarray(v).astype(np.int32)
    q, lv = 2 ** interval, 2 ** (8 - interval) + 1
    strides = (lv ** 3, lv ** 2, lv, 1)
    m = v >> interval
    f = v & (q - 1)
    idx = m[0] * strides[0] + m[1] * strides[1] + m[2] * strides[2] + m[3]
    order = np.argsort(-f, axis=0, kind="stable")
    fs = np.take_along_axis(f, order, axis=0)
    st = np.asarray(strides, dtype=np.int32)[order]
    acc = (q - fs[0])[..., None] * lut[idx]
    for n in range(4):
        idx = idx + st[n]
        wn = fs[n] - (fs[n + 1] if n < 3 else 0)
        acc = acc + wn[..., None] * lut[idx]
    return acc.astype(np.int32)


def lut_interp_numer(lut: np.ndarray, img: np.ndarray, mode: str, r: int, interval: int = 4) -> np.ndarray:
    """One (LUT, mode, rotation) pass over a whole image, unrotated frame.

    img: integer [H, W, C]; returns int32 [H, W, C, oC] = 16 * the value that
    FourSimplexInterpFaster(..., rot=4-r) returns for np.rot90(img, r) padded
    by mode_pad (eval_lut_sr.py:549-564), already rotated back.
    """
    img = np.asarray(img).astype(np.int32)
    H, W = img.shape[:2]
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    vals = []
    for dy, dx in rotated_offsets(mode, r):
        vals.append(img[np.clip(yy + dy, 0, H - 1), np.clip(xx + dx, 0, W - 1)])
    return simplex_numer(lut, np.stack(vals, axis=0), interval)


def _rne_div(n: np.ndarray, d: int) -> np.ndarray:
    """round-half-to-even of n/d for integer n (np.round semantics), exact."""
    n = np.asarray(n).astype(np.int64)
    qq = np.floor_divide(n, d)
    rr = n - qq * d
    up = (2 * rr > d) | ((2 * rr == d) & ((qq & 1) == 1))
    return qq + up


def stage1_feat(img_u8: np.ndarray, luts: dict, modes: str = "sct") -> np.ndarray:
    """feat = rne(clip(pred / len(modes), 0, 255)), uint8 [H, W, C].

    luts: {"s1_<mode>r0": int8 [L**4, 1]}.  eval_lut_sr.py:541-577.
    """
    img = np.asarray(img_u8)
    acc = 0
    for mode in modes:
        lut = np.asarray(luts["s1_{}r0".format(mode)]).reshape(-1, 1)
        for r in range(4):
            acc = acc + lut_interp_numer(lut, img, mode, r)[..., 0]
    feat = np.clip(_rne_div(acc, Q * len(modes)), 0, 255)
    return feat.astype(np.uint8)


def stage2_hyper(feat_u8: np.ndarray, luts: dict, oC: int, modes2: str = "sct") -> np.ndarray:
    """hyper numerators hq = rne(clip(pred/(4*len(modes2)) + 127, 0, 255)), uint8 [H, W, C, oC].

    Rotations 0/2 use LUT ...r0, rotations 1/3 use ...r1 (eval_lut_sr.py:579-628).
    The reference's float hyper map is float32(hq) / 255.
    """
    feat = np.asarray(feat_u8)
    acc = 0
    for mode in modes2:
        for r in range(4):
            lut = np.asarray(luts["s2_{}r{}".format(mode, r & 1)]).reshape(-1, oC)
            acc = acc + lut_interp_numer(lut, feat, mode, r)
    d = Q * 4 * len(modes2)
    hq = np.clip(_rne_div(acc + 127 * d, d), 0, 255)
    return hq.astype(np.uint8)


def lut_stages(img_u8, luts, oC, modes="sct", modes2="sct"):
    feat = stage1_feat(img_u8, luts, modes)
    return feat, stage2_hyper(feat, luts, oC, modes2)


# ----------------------------------------------------------------------------
# SR geometry  (resize_right2d_numpy.py:57-140)
# ----------------------------------------------------------------------------
def out_size(n_in: int, scale: float) -> int:
    return int(math.ceil(scale * n_in))          # :41-45


def sr_axis_tables(n_in: int, n_out: int, scale: float, S: int):
    """1-D tables equivalent to the dense field_of_view / dis maps.

    Returns (left int64 [n_out] in UNPADDED source coordinates, dis float64
    [n_out, S], pad_lo, pad_hi).  Operation order follows the reference
    exactly: g = i/s + (n_in-1)/2 - (n_out-1)/(2s); left = ceil(g - S/2 - eps);
    pad_lo = -left[0]; dis = (g + pad_lo) - (left + pad_lo + a).
    """
    s = float(scale)
    i = np.arange(n_out)
    g = i / s + (n_in - 1) / 2 - (n_out - 1) / (2 * s)
    left = np.ceil(g - S / 2 - EPS32).astype(np.int64)
    pad_lo = int(-left[0])
    pad_hi = int(left[-1] + (S - 1) - n_in + 1)
    gp = g + pad_lo
    fov = (left + pad_lo)[:, None] + np.arange(S)[None, :]
    dis = gp[:, None] - fov
    return left, dis, pad_lo, pad_hi


def sr_axis_tables_torch32(n_in: int, n_out: int, scale: float, S: int):
    """The same tables in the float32 arithmetic of the reference's torch classes
    (Resize2dTorch.get_distance, resize_right2d_torch.py:48-103); dis is returned as float64 values of the float32s."""
    f = np.float32
    g = np.arange(n_out, dtype=np.int64).astype(f) / f(scale)
    g = g + f((n_in - 1) / 2)
    g = g - f((n_out - 1) / (2 * float(scale)))
    left = np.ceil((g - f(S / 2)) - f(EPS32)).astype(np.int64)
    pad_lo = int(-left[0])
    pad_hi = int(left[-1] + (S - 1) - n_in + 1)
    gp = g + f(pad_lo)
    fov = (left + pad_lo)[:, None] + np.arange(S)[None, :]
    dis = (gp[:, None] - fov.astype(f)).astype(f)
    return left, dis.astype(np.float64), pad_lo, pad_hi


def _hyper_f32(hq_u8):
    """float32(hq)/255 exactly as eval_lut_sr.py:623-628."""
    return (np.asarray(hq_u8).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def _gauss_params(hq, max_sigma):
    """rho, sigma_x, sigma_y formed in float32 (resize_right2d_numpy.py:168-170)."""
    h = _hyper_f32(hq)
    rho = h[..., 0] * np.float32(2) - np.float32(1)
    sx = h[..., 1] * np.float32(max_sigma)
    sy = h[..., 2] * np.float32(max_sigma)
    return rho.astype(np.float32), sx.astype(np.float32), sy.astype(np.float32)


def _gauss_w(rho, sx, sy, dx, dy):
    # resize_right2d_numpy.py:150-160 (float32 params promoted by float64 distances)
    xn = (sx * dx) ** 2
    yn = (sy * dy) ** 2
    xyn = sx * dx * sy * dy
    return np.exp(-0.5 * (xn - 2 * rho * xyn + yn))


def _lin_alpha(x, alpha):
    # resize_right2d_numpy.py:233-235
    return (alpha * x + 1) * ((-1 <= x) & (x < 0)) + (1 - alpha * x) * ((0 <= x) & (x <= 1))


def _lin_w(alpha, dx, dy):
    # resize_right2d_numpy.py:237-241
    return np.clip(_lin_alpha(dx, alpha), 0, None) * np.clip(_lin_alpha(dy, alpha), 0, None)


def pad_source_index(i, n, mode):
    """np.pad's source index (and validity) of padded positions i in unpadded coordinates, modes constant / edge /
    reflect / symmetric / wrap -- what np.pad(input, pad_vec, mode=self.pad_mode) (:208, :560) amounts to per tap."""
    i = np.asarray(i)
    inside = (i >= 0) & (i < n)
    if mode == "constant":
        return np.clip(i, 0, n - 1), inside
    if mode in ("edge", "replicate"):
        src = np.clip(i, 0, n - 1)
    elif mode == "reflect":
        p = 2 * (n - 1)
        m = np.mod(i, p) if p else np.zeros_like(i)
        src = np.where(m < n, m, p - m)
    elif mode == "symmetric":
        m = np.mod(i, 2 * n)
        src = np.where(m < n, m, 2 * n - 1 - m)
    elif mode in ("wrap", "circular"):
        src = np.mod(i, n)
    else:
        raise ValueError(mode)
    return src, np.ones_like(inside)


def resize_params_f32(feat, p0, p1, p2, sh, sw, S=2, max_sigma=10, kind="gauss", geometry="f64", pad_mode="constant"):
    """Spatially-varying SR from float32 [C,H,W] maps (the class API of
    SteeringGaussianResize2dNumpy.resize / AmplifiedLinearResize2dNumpy.resize).

    feat: [C,H,W]; p0,p1,p2: hyper maps in [0,1] ([C,H,W]; p1,p2 unused for
    kind="linear").  Returns float64 [C,outH,outW].
    """
    feat = np.asarray(feat, dtype=np.float32)
    C, H, W = feat.shape
    oH, oW = out_size(H, sh), out_size(W, sw)
    tables = sr_axis_tables if geometry == "f64" else sr_axis_tables_torch32       # torch classes: float32 geometry
    aa = 1.0
    if geometry == "f64" and sh < 1.0:
        # anti-aliasing of the numpy classes (resize_right2d_numpy.py:51-55, 186-193): enlarged support for both
        # kinds; the Gaussian additionally sees distances scaled by m.  The reference reads scale_factors[0] (channels,
        # = 1) and [1] (rows) of its [C, H, W] list, so only a ROW factor < 1 triggers it and m = sh.
        m = min(sh, 1.0)
        S = math.ceil(S / m)
        if kind == "gauss":
            aa = m
    lx, disx, _, _ = tables(H, oH, sh, S)
    ly, disy, _, _ = tables(W, oW, sw, S)
    disx, disy = aa * disx, aa * disy
    if kind == "gauss":
        rho = np.asarray(p0, np.float32) * 2 - 1
        sx = np.asarray(p1, np.float32) * max_sigma
        sy = np.asarray(p2, np.float32) * max_sigma
    elif kind == "linear":
        alpha = (np.asarray(p0, np.float32) * 2 - 1)
        alpha = max_sigma * alpha
    # else: fixed kernel of interp_methods.py (Resize2dTorch.resize, resize_right2d_torch.py:105-138), no hyper maps
    num = np.zeros((C, oH, oW), np.float64)
    den = np.zeros((C, oH, oW), np.float64)
    # summation order of the reference: column offset major, row offset minor
    # (numpy meshgrid 'xy' transposes the patch enumeration, :95-98, :200-204)
    for a in range(S):          # column offset
        for b in range(S):      # row offset
            rr = lx + b
            cc = ly + a
            rcl = np.clip(rr, 0, H - 1)
            ccl = np.clip(cc, 0, W - 1)
            rs, okr = pad_source_index(rr, H, pad_mode)           # image operand: np.pad(..., mode=pad_mode)  (:208)
            cs, okc = pad_source_index(cc, W, pad_mode)
            inside = okr[:, None] & okc[None, :]
            dx = disx[:, b][:, None]
            dy = disy[:, a][None, :]
            if kind == "gauss":
                w = _gauss_w(rho[:, rcl][:, :, ccl], sx[:, rcl][:, :, ccl],
                             sy[:, rcl][:, :, ccl], dx[None], dy[None])
            elif kind == "linear":
                w = _lin_w(alpha[:, rcl][:, :, ccl], dx[None], dy[None])
            else:
                w = (fixed_kernel(kind, dx) * fixed_kernel(kind, dy))[None]
            val = feat[:, rs][:, :, cs].astype(np.float64) * inside[None]
            num += w * val
            den += w
    return num / den


def _split_hq(hq, kind):
    h = _hyper_f32(hq)                       # [H,W,C,oC]
    h = np.transpose(h, (3, 2, 0, 1))        # [oC,C,H,W]
    if kind == "gauss":
        return h[0], h[1], h[2]
    return h[0], None, None


def resize_u8(feat_u8, hq_u8, sh, sw, S=2, max_sigma=10, kind="gauss"):
    """stage 3 from the uint8 stage outputs: returns float64 [outH,outW,C]."""
    feat = np.transpose(np.asarray(feat_u8).astype(np.float32), (2, 0, 1))
    p0, p1, p2 = _split_hq(hq_u8, kind)
    out = resize_params_f32(feat, p0, p1, p2, sh, sw, S, max_sigma, kind)
    return np.transpose(out, (1, 2, 0))


def to_u8(x):
    """clip(round(x), 0, 255).astype(uint8)  (eval_lut_sr.py:663-665)."""
    return np.clip(np.round(x), 0, 255).astype(np.uint8)


def sr_pipeline(img_u8, luts, sh, sw, S=2, max_sigma=10, linear=False,
                modes="sct", modes2="sct", return_all=False):
    """End-to-end counterpart of eltr._worker (eval_lut_sr.py:514-665)."""
    oC = 1 if linear else 3
    if linear:               # harness builds AmplifiedLinearResize2dNumpy() with defaults (:482-484)
        S, max_sigma = 2, 1
    feat, hq = lut_stages(img_u8, luts, oC, modes, modes2)
    out = resize_u8(feat, hq, sh, sw, S, max_sigma, "linear" if linear else "gauss")
    if return_all:
        return feat, hq, out, to_u8(out)
    return to_u8(out)


# ----------------------------------------------------------------------------
# homographic warp  (resize_right2d_numpy.py:284-449, 496-636)
# ----------------------------------------------------------------------------
def warp_geometry(matrix, in_hw, out_hw, S):
    """Per-output-pixel geometry in closed form (dense [oH,oW] arrays).

    Returns dict with gx, gy (projected coords + pad_lo, float64), lx, ly
    (left + pad_lo, int64, BEFORE clipping), pad = (plx, phx, ply, phy).
    """
    H, W = in_hw
    oH, oW = out_hw
    Minv = np.linalg.inv(np.asarray(matrix, dtype=np.float64))
    ii, jj = np.meshgrid(np.arange(oH), np.arange(oW), indexing="ij")
    # (x, y, 1) = (col, row, 1) as float32 integers (exact) -> float64  (:321-327)
    pts = np.stack([jj.ravel().astype(np.float32), ii.ravel().astype(np.float32)], axis=-1)
    pts = np.concatenate([pts, np.ones([pts.shape[0], 1])], axis=-1)
    g = np.dot(Minv, pts.transpose(1, 0)).transpose(1, 0)
    g[:, 0] /= g[:, -1]
    g[:, 1] /= g[:, -1]
    gx = g[:, 1].reshape(oH, oW).clip(0, H)     # row coordinate  (:335-339)
    gy = g[:, 0].reshape(oH, oW).clip(0, W)     # col coordinate
    lx = np.int_(np.ceil(gx - S / 2 - EPS32))
    ly = np.int_(np.ceil(gy - S / 2 - EPS32))
    # pad from the corner entries only (:363-369); fov[-1,-1] = left[-1,-1] + S-1
    plx = max(-int(lx[0, 0]), 0)
    phx = max(int(lx[-1, -1]) + S - 1 - H + 1, 0)
    ply = max(-int(ly[0, 0]), 0)
    phy = max(int(ly[-1, -1]) + S - 1 - W + 1, 0)
    return dict(gx=gx + plx, gy=gy + ply, lx=lx + plx, ly=ly + ply, pad=(plx, phx, ply, phy))


def fixed_kernel(kind, x):
    """cubic / lanczos2 / lanczos3 / bilinear 1-D kernels (resize_right/interp_methods.py:35-64)."""
    x = np.asarray(x, dtype=np.float64)
    pi = math.pi
    if kind == "cubic":
        a = np.abs(x)
        a2, a3 = a ** 2, a ** 3
        return (1.5 * a3 - 2.5 * a2 + 1.) * (a <= 1.) + (-0.5 * a3 + 2.5 * a2 - 4. * a + 2.) * ((1. < a) & (a <= 2.))
    if kind == "lanczos2":
        return ((np.sin(pi * x) * np.sin(pi * x / 2) + EPS32) / ((pi ** 2 * x ** 2 / 2) + EPS32)) * (np.abs(x) < 2)
    if kind == "lanczos3":
        return ((np.sin(pi * x) * np.sin(pi * x / 3) + EPS32) / ((pi ** 2 * x ** 2 / 3) + EPS32)) * (np.abs(x) < 3)
    if kind == "bilinear":
        return (x + 1) * ((-1 <= x) & (x < 0)) + (1 - x) * ((0 <= x) & (x <= 1))
    raise ValueError(kind)


def _warp_core(feat, params, matrix, out_hw, S, kind, max_sigma, pad_mode="constant"):
    feat = np.asarray(feat, dtype=np.float32)
    C, H, W = feat.shape
    oH, oW = out_hw
    geo = warp_geometry(matrix, (H, W), (oH, oW), S)
    plx, phx, ply, phy = geo["pad"]
    pad_vec = ((0, 0), (plx, phx), (ply, phy))
    tmp_in = np.pad(feat, pad_vec, mode=pad_mode)              # (:560)
    tmp_p = [np.pad(p, pad_vec, mode="edge") for p in params]
    num = np.zeros((C, oH, oW), np.float64)
    den = np.zeros((C, oH, oW), np.float64)
    for a in range(S):          # column offset (see resize_params_f32)
        for b in range(S):      # row offset
            fx = np.clip(geo["lx"] + b, 0, H - 1)     # indexes the PADDED arrays (:396-398)
            fy = np.clip(geo["ly"] + a, 0, W - 1)
            dx = geo["gx"] - fx
            dy = geo["gy"] - fy
            if kind == "gauss":
                w = _gauss_w(tmp_p[0][:, fx, fy], tmp_p[1][:, fx, fy], tmp_p[2][:, fx, fy], dx[None], dy[None])
            elif kind == "linear":
                w = _lin_w(tmp_p[0][:, fx, fy], dx[None], dy[None])
            elif kind == "nearest":   # box  (interp_methods.py:67-70)
                bx = ((-1 <= dx) & (dx < 0)).astype(np.float64) + ((0 <= dx) & (dx <= 1)).astype(np.float64)
                by = ((-1 <= dy) & (dy < 0)).astype(np.float64) + ((0 <= dy) & (dy <= 1)).astype(np.float64)
                w = np.broadcast_to((bx * by)[None], (C, oH, oW))
            else:                     # fixed kernels (interp_methods.py:35-64, 73-95)
                w = np.broadcast_to((fixed_kernel(kind, dx) * fixed_kernel(kind, dy))[None], (C, oH, oW))
            num += w * tmp_in[:, fx, fy]
            den += w
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den


def warp_params_f32(feat, p0, p1, p2, matrix, out_hw, S=2, max_sigma=10, kind="gauss", pad_mode="constant"):
    """SteeringGaussianWarp2dNumpy.warp / AmplifiedLinearWarp2dNumpy.warp / NearestWarp2dNumpy.warp
    on float32 [C,H,W] maps; float64 [C,oH,oW] out (NaN where all weights vanish)."""
    if kind == "gauss":
        rho = np.asarray(p0, np.float32) * 2 - 1
        sx = np.asarray(p1, np.float32) * max_sigma
        sy = np.asarray(p2, np.float32) * max_sigma
        params = [rho, sx, sy]
    elif kind == "linear":
        alpha = np.asarray(p0, np.float32) * 2 - 1
        params = [max_sigma * alpha]
    else:
        params = []
    return _warp_core(feat, params, matrix, out_hw, S, kind, max_sigma, pad_mode)


def warp_u8(feat_u8, hq_u8, matrix, out_hw, S=2, max_sigma=10, kind="gauss"):
    feat = np.transpose(np.asarray(feat_u8).astype(np.float32), (2, 0, 1))
    p0, p1, p2 = _split_hq(hq_u8, kind)
    out = warp_params_f32(feat, p0, p1, p2, matrix, out_hw, S, max_sigma, kind)
    return np.transpose(out, (1, 2, 0))


def warp_mask(in_hw, matrix, out_hw, border=4, C=3):
    """Validity mask of eval_lut_warp.py:197-204, 229: nearest-warp (S=1, box)
    of a white image with a `border`-px black frame, == 255.  bool [oH,oW,C]."""
    H, W = in_hw
    white = np.zeros((C, H, W), np.float32)
    white[:, border:H - border, border:W - border] = 255
    m = warp_params_f32(white, None, None, None, matrix, out_hw, S=1, kind="nearest")
    return np.transpose(m == 255, (1, 2, 0))


def warp_pipeline(img_u8, luts, matrix, out_hw, S=2, max_sigma=10, linear=False,
                  modes="sct", modes2="sct", return_all=False):
    """End-to-end counterpart of eltr._worker (eval_lut_warp.py:70-222)."""
    oC = 1 if linear else 3
    if linear:
        S, max_sigma = 2, 1
    feat, hq = lut_stages(img_u8, luts, oC, modes, modes2)
    out = warp_u8(feat, hq, matrix, out_hw, S, max_sigma, "linear" if linear else "gauss")
    if return_all:
        return feat, hq, out, to_u8(np.nan_to_num(out, nan=0.0))
    return to_u8(np.nan_to_num(out, nan=0.0))


# ----------------------------------------------------------------------------
# metrics (common/utils.py)
# ----------------------------------------------------------------------------
def rgb2y(img):
    """Y channel of _rgb2ycbcr (common/utils.py:46-76)."""
    T0 = np.array([0.256788235294118, 0.504129411764706, 0.097905882352941])
    return np.dot(np.asarray(img).reshape(-1, 3), T0).reshape(img.shape[:2]) + 16


def psnr_y(gt_u8, out_u8, shave):
    """eval_lut_sr.py:735-742 + common/utils.py:138-151."""
    if gt_u8.shape != out_u8.shape:
        ph, pw = out_u8.shape[:2]
        gt_u8 = gt_u8[:ph, :pw]
        gh, gw = gt_u8.shape[:2]
        out_u8 = out_u8[:gh, :gw]
    a = np.array(rgb2y(gt_u8), dtype=np.float32)
    b = np.array(rgb2y(out_u8), dtype=np.float32)
    diff = b - a
    if shave > 0:
        diff = diff[shave:-shave, shave:-shave]
    rmse = np.sqrt(np.mean(np.power(diff, 2)))
    return float(20 * np.log10(255.0 / rmse))


def ssim_y(gt_u8, out_u8):
    """cal_ssim on the Y planes (eval_lut_sr.py:735-743, common/utils.py:177-206): 11x11 Gaussian window
    (cv2.getGaussianKernel(11, 1.5)), 'valid' 2-D convolution, float64; mean of the SSIM map."""
    if gt_u8.shape != out_u8.shape:
        ph, pw = out_u8.shape[:2]
        gt_u8 = gt_u8[:ph, :pw]
        gh, gw = gt_u8.shape[:2]
        out_u8 = out_u8[:gh, :gw]
    x = np.arange(11) - 5.0
    k = np.exp(-(x * x) / (2 * 1.5 * 1.5))
    k = k / k.sum()
    win = np.outer(k, k)

    def conv_valid(a):
        H, W = a.shape
        o = np.zeros((H - 10, W - 10))
        for i in range(11):
            for j in range(11):
                o += win[i, j] * a[i:i + H - 10, j:j + W - 10]
        return o

    a = np.float64(rgb2y(gt_u8))
    b = np.float64(rgb2y(out_u8))
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = conv_valid(a), conv_valid(b)
    s11 = conv_valid(a * a) - mu1 * mu1
    s22 = conv_valid(b * b) - mu2 * mu2
    s12 = conv_valid(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return float(np.mean(m))


def mpsnr(sr_u8, hr_u8, mask):
    """common/utils.py:168-175 (float32 tensors in the reference)."""
    sr = sr_u8.astype(np.float32)
    hr = hr_u8.astype(np.float32)
    m = mask.astype(np.float32)
    diff = m * (sr - hr) / np.float32(255)
    gain = np.float32(m.size) / m.sum(dtype=np.float32)
    mse = float(gain) * np.mean(diff ** 2, dtype=np.float32)
    return float(-10 * np.log10(mse))


# ----------------------------------------------------------------------------
# fine-tuning twin: SWF2LUT.InterpTorchBatch (resample/model.py:172-385), forward and the autograd gradients
# ----------------------------------------------------------------------------
_SWF_LSB_MODE = {"s": "s", "d": "d", "y": "y", "c": "y", "t": "y"}      # c, t read their LSBs at the 'y' pixels (:229-243)
_SWF_PATTERN = {"s": ((0, 0), (0, 1), (1, 0), (1, 1)), "d": ((0, 0), (0, 2), (2, 0), (2, 2)),
                "y": ((0, 0), (1, 1), (1, 2), (2, 1)), "c": ((0, 0), (0, 1), (0, 2), (0, 3)),
                "t": ((0, 0), (1, 1), (2, 2), (3, 3))}


def swf2lut_interp(weight, outC, mode, img_in, bd, grad_out=None):
    """weight float32 [17^4, outC] (LUT / 127), img_in float32 [B, C, h+bd, w+bd] integer-valued.
    Returns out [B, C*outC, h, w] float32; with grad_out also (grad_weight, grad_img) as torch autograd derives them
    for the reference code: round = straight-through, clamp gate on the rounded value, torch.remainder passes the
    gradient to the LSB source pixels, ties ordered by the reference's case chain (later axis first)."""
    if mode not in _SWF_PATTERN:
        raise ValueError("Mode {} not implemented.".format(mode))
    weight = np.asarray(weight, np.float32)
    img = np.asarray(img_in, np.float32)
    B, Cn, hp, wp = img.shape
    h, w = hp - bd, wp - bd
    rq = np.round(weight * np.float32(127))
    lut = np.clip(rq, -127, 127).astype(np.float32)
    gate = ((rq >= -127) & (rq <= 127))
    pm, pl = _SWF_PATTERN[mode], _SWF_PATTERN[_SWF_LSB_MODE[mode]]
    ii = img.astype(np.int64)
    m = np.stack([ii[:, :, dy:dy + h, dx:dx + w] // Q for dy, dx in pm])            # [4,B,C,h,w]
    f = np.stack([ii[:, :, dy:dy + h, dx:dx + w] % Q for dy, dx in pl])
    strides = np.array([L ** 3, L ** 2, L, 1], np.int64).reshape(4, 1, 1, 1, 1)
    key = f * 4 + np.arange(4).reshape(4, 1, 1, 1, 1)
    order = np.argsort(-key, axis=0, kind="stable")                                  # axis stepped at sorted position n
    fs = np.take_along_axis(f, order, axis=0)
    step = np.take_along_axis(np.broadcast_to(strides, f.shape), order, axis=0)
    idx = [np.sum(m * strides, axis=0)]
    for n in range(4):
        idx.append(idx[-1] + step[n])
    wts = [Q - fs[0], fs[0] - fs[1], fs[1] - fs[2], fs[2] - fs[3], fs[3]]
    P = [lut[i] for i in idx]                                                        # each [B,C,h,w,outC]
    acc = sum(wn[..., None].astype(np.float32) * Pn for wn, Pn in zip(wts, P)) / np.float32(Q)
    out = np.transpose(acc, (0, 1, 4, 2, 3)).reshape(B, Cn * outC, h, w)
    if grad_out is None:
        return out
    g = np.asarray(grad_out, np.float64).reshape(B, Cn, outC, h, w).transpose(0, 1, 3, 4, 2) / Q      # [B,C,h,w,outC]
    gw = np.zeros(weight.shape, np.float64)
    for wn, i in zip(wts, idx):
        np.add.at(gw, i.reshape(-1), (g * wn[..., None]).reshape(-1, outC) * 127.0)
    gw *= gate
    gimg = np.zeros(img.shape, np.float64)
    bb, cc, yy, xx = np.meshgrid(np.arange(B), np.arange(Cn), np.arange(h), np.arange(w), indexing="ij")
    ldy = np.array([p[0] for p in pl]).reshape(4, 1, 1, 1, 1)
    ldx = np.array([p[1] for p in pl]).reshape(4, 1, 1, 1, 1)
    for n in range(4):
        gf = np.sum(g * (P[n + 1].astype(np.float64) - P[n].astype(np.float64)), axis=-1)
        ay = np.take_along_axis(np.broadcast_to(ldy, f.shape), order, axis=0)[n]
        ax = np.take_along_axis(np.broadcast_to(ldx, f.shape), order, axis=0)[n]
        np.add.at(gimg, (bb, cc, yy + ay, xx + ax), gf)
    return out, gw, gimg


def load_luts(model_dir, linear=False, lut_name="LUTft", modes="sct", modes2="sct"):
    """LUT dictionary as eval_lut_sr.py:750-775 builds it (kept int8)."""
    import os
    oC = 1 if linear else 3
    d = {}
    for mode in modes:
        d["s1_{}r0".format(mode)] = np.load(os.path.join(model_dir, "{}_s1_{}r0.npy".format(lut_name, mode))).reshape(-1, 1)
    for mode in modes2:
        for r in (0, 1):
            d["s2_{}r{}".format(mode, r)] = np.load(
                os.path.join(model_dir, "{}_s2_{}r{}.npy".format(lut_name, mode, r))).reshape(-1, oC)
    return d


# ----------------------------------------------------------------------------
# net -> LUT transfer (resample/transfer_to_lut.py:12-170 driving common/network.py:40-163)
# ----------------------------------------------------------------------------
def transfer_inputs(interval: int = 4) -> np.ndarray:
    """The 17^4 x 4 sampled pixel tuples of get_input_tensor (transfer_to_lut.py:12-42), float32 in [0,1]:
    base = 0, 16, ..., 240, 255 (:14-15); column 0 (pixel a) is the slowest axis (:33-37), which is what makes
    index = a*L^3 + b*L^2 + c*L + d in FourSimplexInterpFaster."""
    base = np.arange(0, 257, 2 ** interval)
    base[-1] -= 1
    L = len(base)
    idx = np.stack(np.meshgrid(*([np.arange(L)] * 4), indexing="ij"), axis=-1).reshape(-1, 4)
    return (base[idx].astype(np.float32) / np.float32(255.0)).astype(np.float32)


def srnet_forward(weights: dict, key: str, x: np.ndarray, dtype=np.float64) -> np.ndarray:
    """One SRNet of the reference's SRNetsSWF2 (resample/model.py:81-99) on [N,4] pixel tuples (a,b,c,d):
    SRUnit (network.py:40-71): conv1 over the 4 sampled pixels (2x2 kernel for mode s, 1x4 for c/t after the pixel
    pick of SRNet.forward :139-152 -- either way the flattened kernel meets (a,b,c,d) in order) + ReLU, four dense
    1x1 layers whose outputs are concatenated to their inputs (:26-37), conv6 + tanh.  Returns [N, outC] in (-1, 1).
    `weights`: the module's state_dict as arrays (assets/models/<model>/srnets_weights.npz)."""
    p = key + ".model."
    h = x.astype(dtype)
    w1 = weights[p + "conv1.conv.weight"].reshape(-1, 4).astype(dtype)
    h = np.maximum(h @ w1.T + weights[p + "conv1.conv.bias"].astype(dtype), 0)
    for k in (2, 3, 4, 5):
        w = weights[p + "conv%d.conv1.conv.weight" % k].reshape(64, -1).astype(dtype)
        f = np.maximum(h @ w.T + weights[p + "conv%d.conv1.conv.bias" % k].astype(dtype), 0)
        h = np.concatenate([h, f], axis=1)
    w6 = weights[p + "conv6.conv.weight"].reshape(-1, h.shape[1]).astype(dtype)
    return np.tanh(h @ w6.T + weights[p + "conv6.conv.bias"].astype(dtype))


def transfer_lut(weights: dict, key: str, interval: int = 4, dtype=np.float64, return_float=False):
    """LUT_<key>.npy of transfer_to_lut.py: round(clamp(net(x), -1, 1) * 127) as int8, [17^4, outC] (:117-119)."""
    y = srnet_forward(weights, key, transfer_inputs(interval), dtype)
    lut = np.round(np.clip(y, -1, 1) * 127).astype(np.int8)
    return (lut, y) if return_float else lut
