"""Drop-in for the LUT interpolation function of the reference's
resample/eval_lut_sr.py (and its twin import in resample/eval_lut_warp.py:17)."""
from __future__ import annotations

import numpy as np

from .. import _lib, ops

mode_pad_dict = {"s": 1, "d": 2, "y": 2, "c": 3, "t": 3}


def FourSimplexInterpFaster(weight, img_in, h, w, interval, rot, upscale=4, mode="s", oC=1):
    """4-simplex interpolation of an int8-valued LUT over 4 sampled pixels.

    Same contract as the reference (resample/eval_lut_sr.py:24-470):
    weight [L^4, oC] (integer valued; L = 17 for the shipped interval 4), img_in [C, h+pad, w+pad] (integer valued
    0..255, already rotated and edge-padded by the caller), returns float64
    [C*oC, h', w'] = np.rot90(result, rot, [1, 2]) / 2**interval.
    Accepts numpy arrays (returns numpy) or CUDA tensors (returns a CUDA tensor).
    """
    interval = int(interval)
    if not 1 <= interval <= 7:
        raise ValueError("interval must be 1..7 (q = 2**interval, L = 2**(8-interval)+1, :27-28)")
    dy, dx = _lib.mode_offsets(mode, 0)            # ValueError("Mode x not implemented.")
    torch = _lib.require_gpu()
    from .. import lazy
    pad = mode_pad_dict[mode]
    h, w, rot = int(h), int(w), int(rot)
    if isinstance(img_in, lazy.LazyArray) and img_in.pending and img_in._recipe and img_in._recipe[0] == "rotpad_chw":
        r = _interp_of_rotpad(torch, lazy, weight, img_in, h, w, interval, rot, mode, oC, pad)
        if r is not None:
            return r
    if isinstance(img_in, lazy.DeviceArray):           # a result of an earlier call, still in HBM (lazy.py)
        src, img_in, lazy_out = img_in, img_in.t, True
    else:
        src, lazy_out = None, False
    as_numpy = not isinstance(img_in, torch.Tensor)
    img = _upload_image(torch, np.asarray(img_in)) if as_numpy else img_in
    lut, planes = _device_lut(torch, weight, oC, img.device)
    if img.dtype not in (torch.uint8, torch.float32):
        # float64 images are rounded and clipped in float64 first (a value within float32's epsilon of x.5 must not round twice);
        # integer images are exact in float32 for the 0..255 values of the contract
        img = (img.round().clamp(0, 255) if img.dtype == torch.float64 else img).to(torch.float32)
    if img.shape[1] < h + pad or img.shape[2] < w + pad:
        raise ValueError("img_in must be padded by {} pixels for mode {}".format(pad, mode))
    # one launch: float32 / uint8 pixels in, float64 values out, already rotated back by `rot` quarter turns and divided by
    # q (:464-469) -- the kernel stores through the strides of the rotated view (lerf_lut_interp_ex, ABI 7)
    if lazy_out or (as_numpy and lazy.enabled()):
        # the pass itself is deferred (lazy.LazyArray): `pred += FourSimplexInterpFaster(...)` (:555, :564) then runs it ONCE,
        # adding into pred's planes; any other use runs it into planes of its own, as before.  The operand is a private upload
        # or a DeviceArray (whose writes flush pending readers): what the pass reads cannot change before it runs.
        Cn = img.shape[0]
        oh, ow = (h, w) if rot % 2 == 0 else (w, h)
        return _deferred(torch, lazy, (Cn * oC, oh, ow), [src] if src is not None else [],
                         lambda out, acc: ops.lut_interp(img, h, w, dy, dx, lut, interval, rot=rot, planes=planes, out=out, accumulate=acc),
                         interval, img.device)
    out = ops.lut_interp(img, h, w, dy, dx, lut, interval, rot=rot, out_dtype=torch.float64, planes=planes)
    return out.cpu().numpy() if as_numpy else out


def _deferred(torch, lazy, shape, deps, launch, interval, device):
    """the pass as a lazy.LazyArray: make() runs it into fresh float64 planes; recipe ("interp", run, interval, device) lets
    `pred += result` run it into the sum instead (float64 planes, or the int16 numerators lazy.py keeps while it can)"""
    def run(out, accumulate):
        launch(out, accumulate)
        return True
    return lazy.LazyArray(shape, np.float64, deps, lambda: launch(None, False), ("interp", run, int(interval), device))


def _interp_of_rotpad(torch, lazy, weight, chw, h, w, interval, rot, mode, oC, pad):
    """img_in = np.pad(np.rot90(base, k), ((0, p), (0, p), (0, 0)), mode="edge").transpose((2, 0, 1)), still lazy: the pass over it
    == the pass over `base` itself with the pattern's offsets rotated k times and clamped coordinates (what the tile-fused
    stages do, csrc/lerf_common.h mode_offsets), its result turned by k + rot quarter turns -- no rotated or padded copy exists.
    None: not that case (the caller proceeds on the materialised array)."""
    _, base, k, ph, pw = chw._recipe
    Cn, Hp, Wp = chw.shape
    if (h, w) != (Hp - ph, Wp - pw) or ph < pad or pw < pad or not 1 <= interval <= 7:
        return None
    bt = base.t                                          # [H, W, C]
    if not bt.is_cuda or bt.dim() != 3 or bt.dtype not in (torch.uint8, torch.float32):
        return None
    dy, dx = _lib.mode_offsets(mode, k)
    lut, planes = _device_lut(torch, weight, oC, bt.device)
    H, W = int(bt.shape[0]), int(bt.shape[1])
    img = bt.permute(2, 0, 1)
    turn = (k + rot) % 4
    oh, ow = (H, W) if turn % 2 == 0 else (W, H)
    return _deferred(torch, lazy, (Cn * oC, oh, ow), [base],
                     lambda out, acc: ops.lut_interp(img, H, W, dy, dx, lut, interval, rot=turn, planes=planes, out=out, accumulate=acc),
                     interval, bt.device)


def _upload_image(torch, a):
    """[C, H, W] numpy image -> device.  The call sites hand over `np.pad(...).transpose((2, 0, 1))` (:551-553): a
    transposed VIEW of a contiguous HWC array -- that buffer is uploaded as it lies (no host-side gather) and viewed as
    [C, H, W] on the device."""
    from .. import lazy
    if a.ndim == 3 and not a.flags.c_contiguous and a.transpose(1, 2, 0).flags.c_contiguous:
        return lazy.upload(a.transpose(1, 2, 0)).permute(2, 0, 1)
    return lazy.upload(np.ascontiguousarray(a))


# device copies of the LUT arrays the caller passes again and again (9 per model, 24 calls per image): keyed by the host
# buffer and validated against a private copy of its WHOLE content (np.array_equal on a 1-MB table: 0.09 ms; exact, where
# a sample grid or a checksum would not be) -- an array changed in place, however sparsely (a fine-tuning step, a clamp), is
# uploaded afresh
_LUTS = {}


def _device_lut(torch, weight, oC, device):
    """-> (int8 [L^4, oC] on the device, its plane form for the LDS kernel or None)"""
    def planes_of(lut):
        return ops.lut_planes(lut) if oC > 1 and lut.shape[0] == 17 ** 4 else None
    if isinstance(weight, torch.Tensor):
        lut = weight.to(device).reshape(-1, oC)
        lut = lut.round().to(torch.int8) if lut.dtype != torch.int8 else lut
        return lut, None
    w = np.asarray(weight)
    key = (w.__array_interface__["data"][0], w.shape, w.strides, w.dtype.str, int(oC), str(device))
    hit = _LUTS.get(key)
    if hit is not None and np.array_equal(hit[0], w):
        return hit[1], hit[3]
    lut = torch.from_numpy(np.ascontiguousarray(w)).to(device).reshape(-1, oC)
    lut = lut.round().to(torch.int8) if lut.dtype != torch.int8 else lut
    _LUTS.pop(key, None)
    if len(_LUTS) >= 32:
        _LUTS.pop(next(iter(_LUTS)))
    planes = planes_of(lut)
    _LUTS[key] = (w.copy(), lut, w, planes)             # `w` itself keeps the buffer (and with it the key) alive
    return lut, planes
