#!/usr/bin/env python3
"""The CALLER's side of the drop-in boundary: what `eltr._worker` does around the library calls
(resample/eval_lut_sr.py:541-665), written against nothing but the mirrored names -- FourSimplexInterpFaster,
mode_pad_dict and a resizer object with set_shape / resize -- and plain numpy calls (np.rot90, np.pad, np.round,
np.clip, +=, /).  bench.py --path callsite times it, tests/test_gpu_callsite.py checks its bytes against the
reference's md5s; with `lib = the reference modules` the same function drives the reference itself
(tests/golden/gen_golden.py does that where the fixtures are made).

This is harness code in the reference's protocol, not product code: the product is what it calls.
"""
import numpy as np


def lut_ensemble(interp, pads, luts, img_hwc, stage, modes, out_c, interval, rot_key):
    """sum over patterns x 4 rotations of one LUT pass each (:543-564 stage 1, :580-619 stage 2).
    rot_key(r) -> the 'r0' / 'r1' suffix of the LUT a rotation uses."""
    total = 0
    for m in modes:
        p = pads[m]
        for r in (0, 1, 2, 3):
            lut = luts["s{}_{}{}".format(stage, m, rot_key(r))]
            rot = np.rot90(img_hwc, r)
            h, w, _ = rot.shape
            chw = np.pad(rot, ((0, p), (0, p), (0, 0)), mode="edge").transpose((2, 0, 1))
            total += interp(lut, chw, h, w, interval, 4 - r, upscale=1, mode=m, oC=out_c)
    return total


def worker_sr(interp, pads, resizer, luts, img_lr_hwc_f32, scale_hw, modes="sct", modes2="sct", out_c=3, linear=False,
              interval=4, norm=255):
    """float32 HWC image in -> uint8 HWC image out, by the call sequence of eltr._worker (two-stage models)."""
    # stage 1 (:541-577): every rotation reads the ...r0 table; feat = rne(clip(sum / n_modes))
    s1 = lut_ensemble(interp, pads, luts, img_lr_hwc_f32, 1, modes, 1, interval, lambda r: "r0")
    feat = np.round(np.clip(s1 / len(modes) + 0, 0, norm)).astype(np.float32).transpose((1, 2, 0))
    # stage 2 (:579-628): rotations 0 / 2 read ...r0, 1 / 3 read ...r1; hyper = rne(clip(sum / (4 n) + norm // 2)) / norm
    s2 = lut_ensemble(interp, pads, luts, feat, 2, modes2, out_c, interval, lambda r: "r%d" % (r & 1))
    hyper = np.round(np.clip(s2 / (len(modes2) * 4) + norm // 2, 0, norm)).astype(np.float32) / float(norm)
    # stage 3 (:644-665)
    chw = feat.transpose((2, 0, 1))
    resizer.set_shape(chw.shape, scale_factors=[scale_hw[0], scale_hw[1]])
    if linear:
        out = resizer.resize(chw, hyper)
    else:
        n = hyper.shape[0]
        out = resizer.resize(chw, hyper[list(range(0, n, 3)), :, :], hyper[list(range(1, n + 1, 3)), :, :],
                             hyper[list(range(2, n + 2, 3)), :, :])
    return np.clip(np.round(out).transpose((1, 2, 0)), 0, norm).astype(np.uint8)


def worker_warp(interp, pads, warper, nn_warper, luts, img_lr_hwc_f32, matrix, gt_hw, modes="sct", modes2="sct", out_c=3, linear=False,
                interval=4, norm=255, border=4):
    """float32 HWC image in -> (uint8 HWC image, boolean HWC validity mask), by the call sequence of the warp harness
    (resample/eval_lut_warp.py:100-233): the LUT stages as in worker_sr, then set_shape(matrix) / warp on both resamplers --
    the learned one on (feat, hyper), the nearest one on a white image with a `border`-pixel black frame (:197-204, 229)."""
    s1 = lut_ensemble(interp, pads, luts, img_lr_hwc_f32, 1, modes, 1, interval, lambda r: "r0")
    feat = np.round(np.clip(s1 / len(modes) + 0, 0, norm)).astype(np.float32).transpose((1, 2, 0))
    s2 = lut_ensemble(interp, pads, luts, feat, 2, modes2, out_c, interval, lambda r: "r%d" % (r & 1))
    hyper = np.round(np.clip(s2 / (len(modes2) * 4) + norm // 2, 0, norm)).astype(np.float32) / float(norm)
    chw = feat.transpose((2, 0, 1))
    out_shape = (chw.shape[0], gt_hw[0], gt_hw[1])
    warper.set_shape(chw.shape, matrix, out_shape)
    white = np.zeros(chw.shape, dtype=np.float32)
    h, w = white.shape[-2:]
    white[:, border:h - border, border:w - border] = 255
    nn_warper.set_shape(chw.shape, matrix, out_shape)
    mask_out = nn_warper.warp(white)
    if linear:
        out = warper.warp(chw, hyper)
    else:
        n = hyper.shape[0]
        out = warper.warp(chw, hyper[list(range(0, n, 3)), :, :], hyper[list(range(1, n + 1, 3)), :, :], hyper[list(range(2, n + 2, 3)), :, :])
    out8 = np.clip(np.round(out).transpose((1, 2, 0)), 0, norm).astype(np.uint8)
    mask = np.array(mask_out.transpose((1, 2, 0)) == 255)
    return out8, mask


def mirror_warp_api(linear=False, support=2, max_sigma=10):
    """(interp, pads, learned warper, nearest warper) of the MI355X package (eval_lut_warp.py:37-44 constructs them)"""
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import (AmplifiedLinearWarp2dNumpy, NearestWarp2dNumpy,
                                                                   SteeringGaussianWarp2dNumpy)
    w = AmplifiedLinearWarp2dNumpy() if linear else SteeringGaussianWarp2dNumpy(support_sz=support, max_sigma=max_sigma)
    return FourSimplexInterpFaster, mode_pad_dict, w, NearestWarp2dNumpy()


def mirror_api(linear=False, support=2, max_sigma=10):
    """(interp, pads, resizer) of the MI355X package -- the three names INTEGRATION.md section 2 swaps in"""
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import AmplifiedLinearResize2dNumpy, SteeringGaussianResize2dNumpy
    if linear:
        return FourSimplexInterpFaster, mode_pad_dict, AmplifiedLinearResize2dNumpy()        # defaults, as eval_lut_sr.py:482-484
    return FourSimplexInterpFaster, mode_pad_dict, SteeringGaussianResize2dNumpy(support_sz=support, max_sigma=max_sigma)


def float_luts(luts_i8):
    """the LUT dictionary as the reference's loader holds it (:750-775): float32 [L^4, oC]"""
    return {k: np.ascontiguousarray(v.reshape(v.shape[0], -1).astype(np.float32)) for k, v in luts_i8.items()}
