#!/usr/bin/env python3
"""The CALLER's side of the drop-in boundary: what `eltr._worker` does around the library calls
(resample/eval_lut_sr.py:541-665), written against nothing but the mirrored names -- FourSimplexInterpFaster,
mode_pad_dict and a resizer object with set_shape / resize -- and plain numpy calls (np.rot90, np.pad, np.round,
np.clip, +=, /).  bench.py --path callsite times it, tests/test_gpu_callsite.py checks its bytes against the
reference's md5s; with `lib = the reference modules` the same function drives the reference itself
(tests/golden/gen_golden.py does that where the fixtures are made).

`tail_sr` / `tail_warp` continue to the END of the two workers (resample/eval_lut_sr.py:667-744,
resample/eval_lut_warp.py:221-302): the PNG / .npy files they save and the metrics they return, formed from the library's
results with the same kinds of statements (Image.fromarray, np.save, slicing, np.reshape / np.dot / item assignment of the
colour transform, np.array(..., dtype=float32), torch.Tensor(...), boolean arithmetic with the mask) -- restated here, the
metric helpers included (common/utils.py:46-76, 138-206), because the reference's files do not travel.

This is harness code in the reference's protocol, not product code: the product is what it calls.
"""
import os

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
              interval=4, norm=255, keep=None):
    """float32 HWC image in -> uint8 HWC image out, by the call sequence of eltr._worker (two-stage models).
    `keep` (a dict) receives the intermediates the worker's tail uses: feat_chw, hyper."""
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
    if keep is not None:
        keep.update(feat_chw=chw, hyper=hyper)
    return np.clip(np.round(out).transpose((1, 2, 0)), 0, norm).astype(np.uint8)


def worker_warp(interp, pads, warper, nn_warper, luts, img_lr_hwc_f32, matrix, gt_hw, modes="sct", modes2="sct", out_c=3, linear=False,
                interval=4, norm=255, border=4, keep=None):
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
    if keep is not None:
        keep.update(feat_chw=chw, hyper=hyper, mask_output=mask_out)
    return out8, mask


# ---- the workers' tails: what the caller does with the library's results ---------------------------------------------------
_YCC = np.array([[0.256788235294118, 0.504129411764706, 0.097905882352941],
                 [-0.148223529411765, -0.290992156862745, 0.439215686274510],
                 [0.439215686274510, -0.367788235294118, -0.071427450980392]])
_YCC_OFF = np.array([[16], [128], [128]])


def ycbcr(img):
    """colour transform in the statement pattern of common/utils.py:46-76: flatten, matrix product, the three offsets added by
    item assignment into column views, reshape"""
    flat = np.reshape(img, (img.shape[0] * img.shape[1], img.shape[2]))
    flat = np.dot(flat, np.transpose(_YCC))
    for k in range(3):
        flat[:, k] += _YCC_OFF[k]
    return np.reshape(flat, [img.shape[0], img.shape[1], img.shape[2]])


def psnr(y_true, y_pred, shave):
    """common/utils.py:138-151: float32 copies, difference, border shaved, 20 log10(255 / rmse)"""
    d = np.array(y_pred, dtype=np.float32) - np.array(y_true, dtype=np.float32)
    if shave > 0:
        d = d[shave:-shave, shave:-shave]
    return 20 * np.log10(255. / np.sqrt(np.mean(np.power(d, 2))))


def ssim(img1, img2):
    """common/utils.py:177-206 (the 11-tap sigma-1.5 Gaussian of cv2.getGaussianKernel written out; scipy's 'valid' convolution)"""
    from scipy import signal
    x = np.arange(11) - 5.0
    k = np.exp(-(x * x) / (2 * 1.5 * 1.5))
    k = (k / k.sum()).reshape(-1, 1)
    win = k * k.T
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = np.float64(img1), np.float64(img2)
    ma, mb = signal.convolve2d(a, win, "valid"), signal.convolve2d(b, win, "valid")
    saa = signal.convolve2d(a * a, win, "valid") - ma * ma
    sbb = signal.convolve2d(b * b, win, "valid") - mb * mb
    sab = signal.convolve2d(a * b, win, "valid") - ma * mb
    return np.mean(((2 * ma * mb + c1) * (2 * sab + c2)) / ((ma * ma + mb * mb + c1) * (saa + sbb + c2)))


def masked_psnr(sr, hr, mask, rgb_range=255):
    """common/utils.py:168-175 on torch tensors"""
    import torch
    diff = mask * (sr - hr) / rgb_range
    gain = mask.nelement() / mask.sum()
    return -10 * torch.log10(gain.item() * diff.pow(2).mean())


def tail_sr(img_out, feat_chw, img_hyper, img_gt, scale_hw, result_dir, name, lut_name="LUTft", norm=255):
    """eval_lut_sr.py:667-744 on the worker's values: `img_out` uint8 HWC (the clip(round(.)) of :663-665), `feat_chw` the stage-1
    image as the resizer got it, `img_hyper` the [C * oC, H, W] float32 hyper maps, `img_gt` the uint8 ground truth.
    Saves <name>_<lut>.png, <name>_lr.png, <name>_gt.png, <name>_<lut>_hyper.npy; returns [psnr, ssim]."""
    from PIL import Image
    Image.fromarray(img_out).save(os.path.join(result_dir, "{}_{}.png".format(name, lut_name)))
    feat8 = np.clip(np.round(feat_chw).transpose((1, 2, 0)), 0, norm).astype(np.uint8)
    Image.fromarray(feat8).save(os.path.join(result_dir, "{}_lr.png".format(name)))
    Image.fromarray(img_gt).save(os.path.join(result_dir, "{}_gt.png".format(name)))
    np.save(os.path.join(result_dir, "{}_{}_hyper.npy".format(name, lut_name)), img_hyper)
    if img_gt.shape != img_out.shape:
        ph, pw, _ = img_out.shape
        img_gt = img_gt[:ph, :pw, :]
        gh, gw, _ = img_gt.shape
        img_out = img_out[:gh, :gw, :]
    y_gt, y_out = ycbcr(img_gt)[:, :, 0], ycbcr(img_out)[:, :, 0]
    return [psnr(y_gt, y_out, max(int(scale_hw[0]), int(scale_hw[1]))), ssim(y_gt, y_out)]


def tail_warp(img_out, mask_output_chw, feat_chw, img_gt, result_dir, name, lut_name="LUTft", norm=255):
    """eval_lut_warp.py:221-302: `img_out` uint8 HWC, `mask_output_chw` the nearest warp of the white image as the warper
    returned it ([C, H, W] float64), `img_gt` uint8 HWC.  Saves <name>_lr.png, <name>_mask.png, <name>_<lut>.png (invalid pixels
    white), <name>_gt.png; returns [mpsnr] (a torch scalar, as in the reference)."""
    import torch
    from PIL import Image
    gt_tensor = torch.Tensor(img_gt)
    mask_output = mask_output_chw.transpose((1, 2, 0))
    mask_tensor = torch.Tensor(np.array(mask_output == 255))
    mp = masked_psnr(torch.Tensor(img_out), gt_tensor, mask_tensor)
    feat8 = np.clip(np.round(feat_chw).transpose((1, 2, 0)), 0, norm).astype(np.uint8)
    Image.fromarray(feat8).save(os.path.join(result_dir, "{}_lr.png".format(name)))
    Image.fromarray(np.array((mask_output == 255) * 255).astype(np.uint8)).save(os.path.join(result_dir, "{}_mask.png".format(name)))
    white = (np.ones_like(np.array(img_gt)) * 255).astype(np.uint8)
    img_out = img_out * np.array(mask_output == 255) + np.array(mask_output != 255) * white
    img_gt = np.array(img_gt) * np.array(mask_output == 255) + np.array(mask_output != 255) * white
    Image.fromarray(img_out).save(os.path.join(result_dir, "{}_{}.png".format(name, lut_name)))
    Image.fromarray(img_gt).save(os.path.join(result_dir, "{}_gt.png".format(name)))
    return [mp]


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
