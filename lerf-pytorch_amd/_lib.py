"""ctypes binding of liblerf_hip.so (C ABI: include/lerf_hip.h).

There is no CPU fallback: if the shared library is missing or a device entry
point is called without a GPU, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The in-tree product library.  The package reads NO environment variable for this: diagnostic builds (cycle stamps, A/B
# variants under csrc/build_*/) are selected explicitly by the tools that own them, through use_library() before the first
# call (tools/stamps.py, `bench.py --lib`, which prints the path in its JSON line).
LIB_PATH = os.path.join(_HERE, "liblerf_hip.so")

LERF_MAX_MODES = 5
LERF_LUT_ENTRIES = 83521
LERF_MAX_SUPPORT = 8
LERF_U8, LERF_F32, LERF_F64, LERF_I16 = 0, 1, 2, 3
KIND_GAUSS, KIND_LINEAR, KIND_NEAREST, KIND_CUBIC, KIND_BILINEAR, KIND_LANCZOS2, KIND_LANCZOS3 = range(7)
KINDS = {"gauss": KIND_GAUSS, "linear": KIND_LINEAR, "nearest": KIND_NEAREST, "cubic": KIND_CUBIC,
         "bilinear": KIND_BILINEAR, "lanczos2": KIND_LANCZOS2, "lanczos3": KIND_LANCZOS3}

# image padding rules (include/lerf_hip.h LERF_PAD_*), by their np.pad and F.pad names
PAD_MODES = {"constant": 0, "edge": 1, "replicate": 1, "reflect": 2, "symmetric": 3, "wrap": 4, "circular": 4}
NUMPY_PAD_MODES = ("constant", "edge", "reflect", "symmetric", "wrap")
TORCH_PAD_MODES = ("constant", "replicate", "reflect", "circular")


def pad_mode_code(name, allowed):
    """LERF_PAD_* of a pad_mode the reference's class would hand to np.pad / F.pad.  The index-remapping modes are
    implemented; numpy's statistical ones (linear_ramp, maximum, mean, median, minimum, empty) are not."""
    if name not in allowed:
        raise NotImplementedError("pad_mode {!r} is not implemented (supported: {})".format(name, ", ".join(allowed)))
    return PAD_MODES[name]


EXPORTS = [
    "lerf_abi_version", "lerf_strerror", "lerf_device_count", "lerf_mode_offsets", "lerf_sr_axis_tables", "lerf_sr_axis_tables_f32",
    "lerf_out_size", "lerf_invert3x3", "lerf_warp_pads", "lerf_lut_interp_i16", "lerf_lut_interp", "lerf_lut_interp_ex", "lerf_numer_epilogue_f32", "lerf_fused_lutpack_bytes", "lerf_fused_lutpack_build",
    "lerf_lut_stages_u8",
    "lerf_resize", "lerf_warp", "lerf_sr_fused_workspace_bytes", "lerf_sr_fused_supported", "lerf_sr_fused_u8",
    "lerf_sr_ragged_workspace_bytes", "lerf_sr_fused_ragged_u8", "lerf_stages_ragged_workspace_bytes", "lerf_stages_packed_ragged_u8",
    "lerf_stages_packed_u8", "lerf_unpack_stages", "lerf_warp_packed", "lerf_rect_copy_u8",
    "lerf_warp_tile_boxes", "lerf_warp_fused_supported", "lerf_warp_fused_u8",
    "lerf_metric_y_sse_u8", "lerf_metric_ssim_y_u8", "lerf_metric_masked_sse_u8",
    "lerf_swf2lut_interp_f32", "lerf_swf2lut_interp_bwd_f32", "lerf_resize_bwd_f32",
    "lerf_srnet_weight_floats", "lerf_srnet_to_lut", "lerf_ubench_lds_gather",
]


class LerfError(RuntimeError):
    pass


class Plane(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int), ("sy", C.c_int64), ("sx", C.c_int64), ("sc", C.c_int64)]


class EpiOp(C.Structure):
    _fields_ = [("op", C.c_int), ("a", C.c_double), ("b", C.c_double)]


class Luts(C.Structure):
    _fields_ = [
        ("n_modes1", C.c_int), ("n_modes2", C.c_int),
        ("modes1", C.c_char * LERF_MAX_MODES), ("modes2", C.c_char * LERF_MAX_MODES),
        ("oC", C.c_int),
        ("s1", C.c_void_p * LERF_MAX_MODES),
        ("s2", (C.c_void_p * 2) * LERF_MAX_MODES),
        ("fused_pack", C.c_void_p),
    ]


class SrGeo(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
        ("left_r", C.c_void_p), ("dis_r", C.c_void_p), ("left_c", C.c_void_p), ("dis_c", C.c_void_p),
        ("dis_r64", C.c_void_p), ("dis_c64", C.c_void_p),
        ("pad_mode", C.c_int),
        ("tie_queue_cap", C.c_int),                                   # 0 = default, n > 0 = n entries, < 0 = no queue
        ("roi_y", C.c_int), ("roi_x", C.c_int), ("roi_h", C.c_int), ("roi_w", C.c_int),   # 0-sized = whole frame
        ("flags", C.c_int),                                           # GEO_* bits (ABI 5)
        ("out_row_pitch", C.c_int),                                   # bytes between output rows of lerf_sr_fused_u8, 0 = dense
    ]


GEO_FORCE_GENERAL, GEO_SINGLE_LAUNCH, GEO_INPUT_DEVICE, GEO_INPUT_HOST, GEO_X2_TABLES, GEO_NO_PERSIST = 1, 2, 4, 8, 16, 32
GEO_TILE_ROWS_64, GEO_TILE_ROWS_32, GEO_TILE_ROWS_16 = 64, 128, 256


class SrItem(C.Structure):          # lerf_sr_item_t: one frame of a ragged launch
    _fields_ = [("img", C.c_void_p), ("out", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("geo", SrGeo)]


class StageItem(C.Structure):       # lerf_stage_item_t
    _fields_ = [("img", C.c_void_p), ("packed", C.c_void_p), ("H", C.c_int), ("W", C.c_int)]


class Rect(C.Structure):            # lerf_rect_t
    _fields_ = [("y", C.c_int), ("x", C.c_int), ("h", C.c_int), ("w", C.c_int), ("off", C.c_int64)]


LERF_MAX_RECTS = 8


class WarpGeo(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
        ("minv", C.c_double * 9),
        ("pad_r_lo", C.c_int), ("pad_r_hi", C.c_int), ("pad_c_lo", C.c_int), ("pad_c_hi", C.c_int),
        ("pad_mode", C.c_int),
        ("out_y0", C.c_int), ("out_x0", C.c_int), ("src_y0", C.c_int),       # ABI 7: a rectangle of the output from a band of the source
    ]


_lib = None


def use_library(path):
    """tools only: load another build of liblerf_hip.so (same ABI) instead of the product library.  Must come before the first
    library call of the process; the choice is visible as `_lib.LIB_PATH`."""
    global LIB_PATH
    if _lib is not None:
        raise LerfError("use_library(%s): liblerf_hip.so is already loaded from %s" % (path, LIB_PATH))
    LIB_PATH = os.path.abspath(path)


def lib():
    """Load liblerf_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LerfError(
            "liblerf_hip.so not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    # ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so and liblerf_hip.so depends on the same
    # SONAME, so whichever is loaded first serves both.  With liblerf_hip.so first, torch ends up on the system runtime it
    # was not built for and the first kernel launch fails with "no ROCm-capable device" (seen when a numpy-class test ran
    # before anything had imported torch); torch first, and both use torch's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.lerf_abi_version.restype = C.c_int
    L.lerf_strerror.restype = C.c_char_p
    L.lerf_strerror.argtypes = [C.c_int]
    L.lerf_device_count.restype = C.c_int
    L.lerf_mode_offsets.argtypes = [C.c_char, C.c_int, C.c_void_p, C.c_void_p]
    L.lerf_sr_axis_tables.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_sr_axis_tables_f32.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_out_size.argtypes = [C.c_int, C.c_double]
    L.lerf_invert3x3.argtypes = [C.c_void_p, C.c_void_p]
    L.lerf_warp_pads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lerf_lut_interp_i16.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.lerf_lut_interp.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Plane), C.c_void_p]
    L.lerf_lut_interp_ex.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Plane), C.c_int, C.c_void_p]
    L.lerf_numer_epilogue_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.lerf_warp_tile_boxes.argtypes = [C.POINTER(WarpGeo), C.c_int, C.c_int, C.c_void_p]
    L.lerf_warp_fused_supported.argtypes = [C.c_int, C.POINTER(Luts), C.POINTER(WarpGeo), C.c_int, C.c_int, C.c_int, C.c_double]
    L.lerf_warp_fused_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Luts), C.POINTER(WarpGeo), C.c_void_p,
                                     C.c_int, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    L.lerf_fused_lutpack_bytes.restype = C.c_size_t
    L.lerf_fused_lutpack_bytes.argtypes = [C.POINTER(Luts)]
    L.lerf_fused_lutpack_build.argtypes = [C.POINTER(Luts), C.c_void_p, C.c_void_p]
    L.lerf_lut_stages_u8.argtypes = [C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.POINTER(Luts),
                                     C.POINTER(Plane), C.POINTER(Plane), C.c_void_p]
    L.lerf_resize.argtypes = [C.POINTER(Plane), C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.POINTER(SrGeo),
                              C.c_int, C.c_double, C.POINTER(Plane), C.c_void_p]
    L.lerf_warp.argtypes = [C.POINTER(Plane), C.POINTER(Plane), C.c_int, C.c_int, C.c_int, C.POINTER(WarpGeo),
                            C.c_int, C.c_double, C.POINTER(Plane), C.c_void_p]
    L.lerf_stages_packed_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Luts),
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    L.lerf_unpack_stages.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_warp_packed.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(WarpGeo), C.c_int, C.c_double,
                                   C.POINTER(Plane), C.c_int64, C.c_void_p]
    L.lerf_rect_copy_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Rect), C.c_int, C.c_int,
                                    C.c_void_p]
    L.lerf_sr_fused_supported.argtypes = [C.c_int, C.POINTER(Luts), C.POINTER(SrGeo), C.c_int, C.c_int, C.c_int, C.c_double]
    L.lerf_sr_ragged_workspace_bytes.restype = C.c_size_t
    L.lerf_sr_ragged_workspace_bytes.argtypes = [C.POINTER(SrItem), C.c_int, C.c_int]
    L.lerf_sr_fused_ragged_u8.argtypes = [C.POINTER(SrItem), C.c_int, C.c_int, C.POINTER(Luts), C.c_int, C.c_double, C.c_void_p,
                                          C.c_size_t, C.c_void_p]
    L.lerf_stages_ragged_workspace_bytes.restype = C.c_size_t
    L.lerf_stages_ragged_workspace_bytes.argtypes = [C.POINTER(StageItem), C.c_int, C.c_int]
    L.lerf_stages_packed_ragged_u8.argtypes = [C.POINTER(StageItem), C.c_int, C.c_int, C.POINTER(Luts), C.c_void_p, C.c_size_t,
                                               C.c_void_p]
    L.lerf_sr_fused_workspace_bytes.restype = C.c_size_t
    L.lerf_sr_fused_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.lerf_sr_fused_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Luts),
                                   C.POINTER(SrGeo), C.c_int, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
    L.lerf_metric_y_sse_u8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]
    L.lerf_metric_ssim_y_u8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.lerf_metric_masked_sse_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.lerf_swf2lut_interp_f32.argtypes = [C.c_void_p, C.c_int, C.c_char, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]
    L.lerf_swf2lut_interp_bwd_f32.argtypes = [C.c_void_p, C.c_int, C.c_char, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_resize_bwd_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SrGeo),
                                      C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_srnet_weight_floats.restype = C.c_size_t
    L.lerf_srnet_weight_floats.argtypes = [C.c_int]
    L.lerf_srnet_to_lut.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_ubench_lds_gather.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for name in EXPORTS:          # AttributeError here = the .so does not match include/lerf_hip.h
        getattr(L, name)
    if L.lerf_abi_version() != 7:
        raise LerfError("liblerf_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().lerf_strerror(rc).decode()
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise LerfError("%s: %s (code %d)" % (what, msg, rc))


# ------------------------------------------------------------------ host-side helpers (no GPU needed)
_MODE_OFFSETS = {}


def mode_offsets(mode: str, rot: int):
    """(dy[4], dx[4]) int8 of a sampling pattern rotated `rot` quarter turns (lerf_mode_offsets); cached: a pure function"""
    hit = _MODE_OFFSETS.get((mode, int(rot)))
    if hit is not None:
        return hit[0].copy(), hit[1].copy()
    dy, dx = _mode_offsets(mode, rot)
    _MODE_OFFSETS[(mode, int(rot))] = (dy.copy(), dx.copy())
    return dy, dx


def _mode_offsets(mode: str, rot: int):
    dy = np.zeros(4, np.int8)
    dx = np.zeros(4, np.int8)
    rc = lib().lerf_mode_offsets(mode.encode()[:1] if mode else b"\0", int(rot), dy.ctypes.data, dx.ctypes.data)
    if rc != 0:
        raise ValueError("Mode {} not implemented.".format(mode))     # resample/eval_lut_sr.py:84
    return dy, dx


def out_size(n_in: int, scale: float) -> int:
    return int(lib().lerf_out_size(int(n_in), float(scale)))


def sr_axis_tables(n_in: int, n_out: int, scale: float, S: int):
    left = np.zeros(n_out, np.int32)
    dis64 = np.zeros((n_out, S), np.float64)
    dis32 = np.zeros((n_out, S), np.float32)
    pads = np.zeros(2, np.int32)
    check(lib().lerf_sr_axis_tables(int(n_in), int(n_out), float(scale), int(S), left.ctypes.data,
                                    dis64.ctypes.data, dis32.ctypes.data, pads.ctypes.data), "lerf_sr_axis_tables")
    return left, dis64, dis32, (int(pads[0]), int(pads[1]))


def sr_axis_tables_f32(n_in: int, n_out: int, scale: float, S: int):
    """float32 tables of the reference's torch classes (resize_right2d_torch.py:48-103)."""
    left = np.zeros(n_out, np.int32)
    dis32 = np.zeros((n_out, S), np.float32)
    pads = np.zeros(2, np.int32)
    check(lib().lerf_sr_axis_tables_f32(int(n_in), int(n_out), float(scale), int(S), left.ctypes.data,
                                        dis32.ctypes.data, pads.ctypes.data), "lerf_sr_axis_tables_f32")
    return left, dis32.astype(np.float64), dis32, (int(pads[0]), int(pads[1]))


def warp_pads(minv: np.ndarray, in_hw, out_hw, S: int):
    """pads (r_lo, r_hi, c_lo, c_hi) for the INVERSE homography (np.linalg.inv of the matrix)."""
    m = np.ascontiguousarray(minv, dtype=np.float64).reshape(9)
    pads = np.zeros(4, np.int32)
    check(lib().lerf_warp_pads(m.ctypes.data, int(in_hw[0]), int(in_hw[1]), int(out_hw[0]), int(out_hw[1]), int(S),
                               pads.ctypes.data), "lerf_warp_pads")
    return tuple(int(p) for p in pads)


# ------------------------------------------------------------------ device plumbing
_GPU_SEEN = None


def require_gpu():
    global _GPU_SEEN
    if _GPU_SEEN is not None:                       # a GPU that was there stays there: the hot call sites ask 200 times per frame
        return _GPU_SEEN
    import torch
    if not torch.cuda.is_available():
        raise LerfError("lerf-pytorch_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                        "and there is no CPU fallback.")
    _GPU_SEEN = torch
    return torch


_TORCH_DT = None


def _dt(t):
    global _TORCH_DT
    import torch
    if _TORCH_DT is None:
        _TORCH_DT = {torch.uint8: LERF_U8, torch.float32: LERF_F32, torch.float64: LERF_F64, torch.int16: LERF_I16}
    return _TORCH_DT[t.dtype]


def plane(t, sy, sx, sc, offset=0):
    """Plane descriptor of torch tensor `t` (element strides)."""
    return Plane(t.data_ptr() + offset * t.element_size(), _dt(t), int(sy), int(sx), int(sc))


def current_stream(device=None):
    """torch's current stream of `device` (default: the current device) as a hipStream_t"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # the handle without a Stream object (7 us less per launch)
    if raw is not None:
        if device is None:
            idx = torch.cuda.current_device()
        else:
            idx = torch.device(device).index
            idx = torch.cuda.current_device() if idx is None else idx
        return C.c_void_p(raw(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class on_device:
    """`with on_device(t):` -- the tensor's device is the current one for the launches inside (the library resolves
    per-device kernel attributes with hipGetDevice, and current_stream() then is the stream of THAT device); no-op for
    host tensors (pinned frames of stream.StreamingSR run on the caller's current device)."""

    def __init__(self, t):
        import torch
        self._g = None
        if getattr(t, "is_cuda", False) and t.device.index != torch.cuda.current_device():    # already current: nothing to switch
            self._g = torch.cuda.device(t.device)

    def __enter__(self):
        if self._g is not None:
            self._g.__enter__()
        return self

    def __exit__(self, *exc):
        if self._g is not None:
            return self._g.__exit__(*exc)
        return False
