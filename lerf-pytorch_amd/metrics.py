"""Evaluation metrics of the reference harness, computed on the device.

    psnr_y(gt, out, shave)   common/utils.py:138-151 PSNR on _rgb2ycbcr(.)[:, :, 0]   (eval_lut_sr.py:741-742)
    ssim_y(gt, out)          common/utils.py:177-206 cal_ssim on the same Y planes      (eval_lut_sr.py:743)
    mpsnr(sr, hr, mask)      common/utils.py:168-175 mPSNR                              (eval_lut_warp.py:233)

Inputs are uint8 HWC RGB frames (numpy arrays or CUDA tensors).  The sums are formed by liblerf_hip.so
(lerf_metric_*); only the two resulting doubles come back to the host.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib


def _dev_u8(a, device=None):
    torch = _lib.require_gpu()
    if isinstance(a, torch.Tensor):
        t = a
    else:
        a = np.asarray(a)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    if t.dtype != torch.uint8:
        raise ValueError("metrics take uint8 (or bool mask) frames")
    return t.to(device or "cuda")


def _crop_pair(gt, out):
    """eval_lut_sr.py:735-739: crop both to the common top-left region."""
    if tuple(gt.shape) != tuple(out.shape):
        ph, pw = out.shape[:2]
        gt = gt[:ph, :pw]
        gh, gw = gt.shape[:2]
        out = out[:gh, :gw]
    return gt, out


def _hwc3(t, name):
    if t.dim() != 3 or t.shape[2] != 3 or t.stride(2) != 1 or t.stride(1) != 3:
        raise ValueError("%s must be an HWC RGB frame with packed pixels" % name)


def _result(torch, device):
    return torch.empty(2, dtype=torch.float64, device=device)


def y_sse(gt, out, shave):
    """(sum of squared float32 Y differences, pixel count) as a device tensor of two float64."""
    torch = _lib.require_gpu()
    gt = _dev_u8(gt)
    out = _dev_u8(out, gt.device)
    gt, out = _crop_pair(gt, out)
    _hwc3(gt, "gt")
    _hwc3(out, "out")
    H, W = int(gt.shape[0]), int(gt.shape[1])
    res = _result(torch, gt.device)
    _lib.check(_lib.lib().lerf_metric_y_sse_u8(C.c_void_p(gt.data_ptr()), gt.stride(0), C.c_void_p(out.data_ptr()),
                                               out.stride(0), H, W, int(shave), C.c_void_p(res.data_ptr()),
                                               _lib.current_stream()), "lerf_metric_y_sse_u8")
    return res


def psnr_y(gt, out, shave):
    sse, n = y_sse(gt, out, shave).tolist()
    return 20.0 * math.log10(255.0 / math.sqrt(sse / n))


def ssim_y(gt, out):
    torch = _lib.require_gpu()
    gt = _dev_u8(gt)
    out = _dev_u8(out, gt.device)
    gt, out = _crop_pair(gt, out)
    _hwc3(gt, "gt")
    _hwc3(out, "out")
    res = _result(torch, gt.device)
    _lib.check(_lib.lib().lerf_metric_ssim_y_u8(C.c_void_p(gt.data_ptr()), gt.stride(0), C.c_void_p(out.data_ptr()),
                                                out.stride(0), int(gt.shape[0]), int(gt.shape[1]),
                                                C.c_void_p(res.data_ptr()), _lib.current_stream()), "lerf_metric_ssim_y_u8")
    s, n = res.tolist()
    return s / n


def mpsnr(sr, hr, mask):
    torch = _lib.require_gpu()
    sr = _dev_u8(sr).contiguous()
    hr = _dev_u8(hr, sr.device).contiguous()
    mask = _dev_u8(mask, sr.device).contiguous()
    if sr.shape != hr.shape or sr.shape != mask.shape:
        raise ValueError("mpsnr: sr, hr and mask must have one shape")
    res = _result(torch, sr.device)
    _lib.check(_lib.lib().lerf_metric_masked_sse_u8(C.c_void_p(sr.data_ptr()), C.c_void_p(hr.data_ptr()),
                                                    C.c_void_p(mask.data_ptr()), sr.numel(), C.c_void_p(res.data_ptr()),
                                                    _lib.current_stream()), "lerf_metric_masked_sse_u8")
    sse, msum = res.tolist()
    n = float(sr.numel())
    # gain = nelement / mask.sum(); mse = gain * mean(diff^2)
    return -10.0 * math.log10((n / msum) * (sse / n))
