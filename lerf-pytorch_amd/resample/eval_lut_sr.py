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
    as_numpy = not isinstance(img_in, torch.Tensor)
    if as_numpy:
        img = torch.from_numpy(np.ascontiguousarray(np.asarray(img_in))).cuda()
        lut = torch.from_numpy(np.ascontiguousarray(np.asarray(weight))).cuda()
    else:
        img, lut = img_in, weight.to(img_in.device)
    img = img.round().clamp(0, 255).to(torch.uint8) if img.dtype != torch.uint8 else img
    lut = lut.reshape(-1, oC)
    lut = lut.round().to(torch.int8) if lut.dtype != torch.int8 else lut
    pad = mode_pad_dict[mode]
    if img.shape[1] < h + pad or img.shape[2] < w + pad:
        raise ValueError("img_in must be padded by {} pixels for mode {}".format(pad, mode))
    num = ops.lut_interp_i16(img, h, w, dy, dx, lut, interval)    # [C, oC, h, w] int16, value * 2^interval
    Cn = img.shape[0]
    out = num.reshape(Cn * oC, h, w)
    out = torch.rot90(out, int(rot), [1, 2]).to(torch.float64) / float(2 ** interval)
    return out.cpu().numpy() if as_numpy else out
