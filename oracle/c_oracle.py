"""ctypes wrapper of oracle/liblerf_oracle.so (the C restatement of the oracle).

TEST INFRASTRUCTURE ONLY -- see lerf_oracle.c.  Used by tests/ for sizes the
numpy oracle is too slow for, and by bench.py's cpu_baseline leg ("port").
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("LERF_ORACLE_LIB") or os.path.join(_HERE, "liblerf_oracle.so")     # (the sanitizer build: oracle/Makefile `asan`)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(LIB)
    return _lib


def threads():
    return int(lib().lerf_oracle_threads())


def set_threads(n):
    """n > 0: that many OpenMP threads for the following calls; n <= 0: one per host core again."""
    lib().lerf_oracle_set_threads(int(n))


def _lut_ptrs(luts, modes, modes2, oC):
    keep = []
    s1 = (C.c_void_p * len(modes))()
    for i, m in enumerate(modes):
        a = np.ascontiguousarray(np.asarray(luts["s1_%sr0" % m]).astype(np.int8).reshape(-1))
        keep.append(a)
        s1[i] = a.ctypes.data
    s2 = (C.c_void_p * (2 * len(modes2)))()
    for i, m in enumerate(modes2):
        for r in (0, 1):
            a = np.ascontiguousarray(np.asarray(luts["s2_%sr%d" % (m, r)]).astype(np.int8).reshape(-1, oC))
            keep.append(a)
            s2[2 * i + r] = a.ctypes.data
    return s1, s2, keep


def lut_stages(img_u8, luts, oC, modes="sct", modes2="sct"):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    H, W, Cn = img.shape
    s1, s2, keep = _lut_ptrs(luts, modes, modes2, oC)
    feat = np.empty((H, W, Cn), np.uint8)
    hq = np.empty((H, W, Cn, oC), np.uint8)
    rc = lib().lerf_oracle_lut_stages(C.c_void_p(img.ctypes.data), H, W, Cn, modes.encode(), len(modes), s1,
                                      modes2.encode(), len(modes2), s2, oC, C.c_void_p(feat.ctypes.data),
                                      C.c_void_p(hq.ctypes.data))
    if rc:
        raise ValueError("lerf_oracle_lut_stages failed")
    return feat, hq


def resize(feat_u8, hq_u8, sh, sw, S=2, max_sigma=10.0, kind="gauss"):
    feat = np.ascontiguousarray(feat_u8, dtype=np.uint8)
    hq = np.ascontiguousarray(hq_u8, dtype=np.uint8)
    H, W, Cn = feat.shape
    oC = hq.shape[3]
    oH, oW = int(np.ceil(sh * H)), int(np.ceil(sw * W))
    out = np.empty((oH, oW, Cn), np.float64)
    rc = lib().lerf_oracle_resize(C.c_void_p(feat.ctypes.data), C.c_void_p(hq.ctypes.data), H, W, Cn, oC,
                                  C.c_double(sh), C.c_double(sw), int(S), C.c_double(max_sigma),
                                  0 if kind == "gauss" else 1, C.c_void_p(out.ctypes.data))
    if rc:
        raise ValueError("lerf_oracle_resize failed")
    return out


_SCRATCH = {}


def sr_u8(img_u8, luts, sh, sw, S=2, max_sigma=10.0, linear=False, modes="sct", modes2="sct", out=None):
    """uint8 HWC -> uint8 HWC.  The work area (stage outputs, geometry tables) is caller-owned in the C API and cached here
    per shape, so a timing loop does not pay a quarter of a gigabyte of malloc + first-touch page faults per frame;
    out: optional uint8 [oH,oW,C] array to write into (same reason)."""
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    H, W, Cn = img.shape
    oC = 1 if linear else 3
    if linear:
        S, max_sigma = 2, 1.0
    s1, s2, keep = _lut_ptrs(luts, modes, modes2, oC)
    oH, oW = int(np.ceil(sh * H)), int(np.ceil(sw * W))
    if out is None:
        out = np.empty((oH, oW, Cn), np.uint8)
    elif out.shape != (oH, oW, Cn) or out.dtype != np.uint8 or not out.flags.c_contiguous:
        raise ValueError("out must be a contiguous uint8 [oH,oW,C] array")
    L = lib()
    L.lerf_oracle_sr_scratch_bytes.restype = C.c_size_t
    nb = int(L.lerf_oracle_sr_scratch_bytes(H, W, Cn, oC, C.c_double(sh), C.c_double(sw), int(S)))
    key = (H, W, Cn, oC, float(sh), float(sw), int(S))
    ws = _SCRATCH.get(key)
    if ws is None or ws.nbytes < nb:
        _SCRATCH.clear()                                          # one shape at a time
        ws = _SCRATCH[key] = np.empty(nb, np.uint8)
    rc = L.lerf_oracle_sr_u8_ws(C.c_void_p(img.ctypes.data), H, W, Cn, modes.encode(), len(modes), s1,
                                modes2.encode(), len(modes2), s2, oC, C.c_double(sh), C.c_double(sw), int(S),
                                C.c_double(max_sigma), 1 if linear else 0, C.c_void_p(out.ctypes.data),
                                C.c_void_p(ws.ctypes.data))
    if rc:
        raise ValueError("lerf_oracle_sr_u8 failed")
    return out


def warp(feat_u8, hq_u8, matrix, out_hw, S=2, max_sigma=10.0, kind="gauss"):
    """float64 [oH,oW,C] (NaN where every weight vanishes); kind in gauss / linear / nearest."""
    feat = np.ascontiguousarray(feat_u8, dtype=np.uint8)
    H, W, Cn = feat.shape
    k = {"gauss": 0, "linear": 1, "nearest": 2}[kind]
    if k == 2:
        hq, oC, hp = None, 0, None
    else:
        hq = np.ascontiguousarray(hq_u8, dtype=np.uint8)
        oC, hp = hq.shape[3], C.c_void_p(hq.ctypes.data)
    minv = np.ascontiguousarray(np.linalg.inv(np.asarray(matrix, dtype=np.float64)))
    out = np.empty((int(out_hw[0]), int(out_hw[1]), Cn), np.float64)
    rc = lib().lerf_oracle_warp(C.c_void_p(feat.ctypes.data), hp, H, W, Cn, oC, C.c_void_p(minv.ctypes.data),
                                int(out_hw[0]), int(out_hw[1]), int(S), C.c_double(max_sigma), k,
                                C.c_void_p(out.ctypes.data))
    if rc:
        raise ValueError("lerf_oracle_warp failed")
    return out


def warp_pads(matrix, in_hw, out_hw, S):
    minv = np.ascontiguousarray(np.linalg.inv(np.asarray(matrix, dtype=np.float64)))
    pads = (C.c_int * 4)()
    lib().lerf_oracle_warp_pads(C.c_void_p(minv.ctypes.data), int(in_hw[0]), int(in_hw[1]), int(out_hw[0]), int(out_hw[1]),
                                int(S), pads)
    return tuple(int(v) for v in pads)


def warp_u8(img_u8, luts, matrix, out_hw, S=2, max_sigma=10.0, linear=False, border=4, modes="sct", modes2="sct",
            with_mask=True):
    """uint8 HWC -> (uint8 [oH,oW,C], bool mask [oH,oW,C] or None): the body of eltr._worker in eval_lut_warp.py."""
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    H, W, Cn = img.shape
    oC = 1 if linear else 3
    if linear:
        S, max_sigma = 2, 1.0
    s1, s2, keep = _lut_ptrs(luts, modes, modes2, oC)
    minv = np.ascontiguousarray(np.linalg.inv(np.asarray(matrix, dtype=np.float64)))
    oH, oW = int(out_hw[0]), int(out_hw[1])
    out = np.empty((oH, oW, Cn), np.uint8)
    mask = np.empty((oH, oW, Cn), np.uint8) if with_mask else None
    rc = lib().lerf_oracle_warp_u8(C.c_void_p(img.ctypes.data), H, W, Cn, modes.encode(), len(modes), s1,
                                   modes2.encode(), len(modes2), s2, oC, C.c_void_p(minv.ctypes.data), oH, oW, int(S),
                                   C.c_double(max_sigma), 1 if linear else 0, int(border), C.c_void_p(out.ctypes.data),
                                   C.c_void_p(mask.ctypes.data) if with_mask else None)
    if rc:
        raise ValueError("lerf_oracle_warp_u8 failed")
    return out, (mask.astype(bool) if with_mask else None)
