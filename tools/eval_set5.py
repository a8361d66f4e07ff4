#!/usr/bin/env python3
"""Set5 tables of the reference's scripts.sh (:33-47) from the MI355X path: the counterpart of running
resample/eval_lut_sr.py and resample/eval_lut_warp.py (Y-PSNR with shave=scale; masked mPSNR).
Metrics follow common/utils.py:46-76, 138-151, 168-175 (computed on the host with numpy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import json
import numpy as np
from PIL import Image
import lerf_pytorch_amd as L

DATA = os.path.join(ROOT, "tests", "data", "Set5")
NAMES = ["baby", "bird", "butterfly", "head", "woman"]


def rgb2y(img):
    t = np.array([0.256788235294118, 0.504129411764706, 0.097905882352941])
    return np.dot(img.reshape(-1, 3), t).reshape(img.shape[:2]) + 16


def psnr_y(gt, out, shave):
    if gt.shape != out.shape:
        gt = gt[:out.shape[0], :out.shape[1]]
        out = out[:gt.shape[0], :gt.shape[1]]
    d = (np.array(rgb2y(out), np.float32) - np.array(rgb2y(gt), np.float32))[shave:-shave, shave:-shave]
    return float(20 * np.log10(255.0 / np.sqrt(np.mean(np.power(d, 2)))))


def mpsnr(sr, hr, mask):
    m = mask.astype(np.float32)
    diff = m * (sr.astype(np.float32) - hr.astype(np.float32)) / np.float32(255)
    return float(-10 * np.log10(float(np.float32(m.size) / m.sum(dtype=np.float32)) * np.mean(diff ** 2, dtype=np.float32)))


def main():
    mats = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_set5.json")))["warp"]
    print("model   | SR x2 / x3 / x4 (Y-PSNR)        | warp isc / osc (mPSNR) | scripts.sh")
    expect = {"lerf-g": "35.71 32.02 30.15 | 33.81 27.89", "lerf-l": "34.84 30.72 29.13 | 32.90 27.13"}
    for model in ("lerf-l", "lerf-g"):
        eng = L.LerfEngine.shipped(model)
        sr = []
        for s in (2, 3, 4):
            ps = []
            for n in NAMES:
                lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (s, s), n + ".png")))
                gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
                ps.append(psnr_y(gt, eng.sr(lr, s), s))
            sr.append(np.mean(ps))
        wp = []
        for p in ("isc", "osc"):
            ms = []
            for n in NAMES:
                lr = np.array(Image.open(os.path.join(DATA, p, n + ".png")))
                gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
                o, mask = eng.warp(lr, np.array(mats["%s/%s/%s" % (model, p, n)]["matrix"]), gt.shape[:2])
                ms.append(mpsnr(o, gt, mask))
            wp.append(np.mean(ms))
        print("%-7s | %.2f / %.2f / %.2f             | %.2f / %.2f          | %s" % (model, sr[0], sr[1], sr[2], wp[0], wp[1], expect[model]))


if __name__ == "__main__":
    main()
