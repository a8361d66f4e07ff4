#!/usr/bin/env python3
"""Throughput of the fused kernel for inputs with different LUT-address statistics (same shapes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench
eng = L.LerfEngine.shipped("lerf-g")
geo = eng.sr_geometry((bench.H, bench.W), 2)
B = 8
def run(name, frames):
    x = torch.from_numpy(frames).cuda()
    out = torch.empty((B, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0, out=out)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 10
    for _ in range(n): ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("%-28s %.3f ms/launch  %.1f Mpix/s" % (name, dt * 1e3, B * 2160 * 3840 / dt / 1e6))
run("noise", bench.synth_frames("noise", B, 1))
run("natural-like", bench.synth_frames("natural", B, 1))
run("constant 128", np.full((B, 1080, 1920, 3), 128, np.uint8))
g = np.tile((np.arange(1920) * 255 // 1919).astype(np.uint8)[None, None, :, None], (B, 1080, 1, 3))
run("horizontal ramp", np.ascontiguousarray(g))
from PIL import Image
def photo():
    names = ["baby", "bird", "butterfly", "head", "woman"]
    ims = [np.array(Image.open(os.path.join(ROOT, "tests/data/Set5/HR", n + ".png"))) for n in names]
    canvas = np.zeros((1080, 1920, 3), np.uint8)
    y = 0; k = 0
    while y < 1080:
        x = 0; rowh = 0
        while x < 1920:
            im = ims[k % 5]; k += 1
            h = min(im.shape[0], 1080 - y); w = min(im.shape[1], 1920 - x)
            canvas[y:y + h, x:x + w] = im[:h, :w]; x += w; rowh = max(rowh, h)
        y += rowh
    return canvas
ph = photo()
run("Set5 HR mosaic (real photos)", np.ascontiguousarray(np.stack([np.roll(ph, 37 * i, axis=1) for i in range(B)])))
rng = np.random.default_rng(0)
run("noise in [96,160)", rng.integers(96, 160, (B, 1080, 1920, 3), dtype=np.uint8))
run("noise, multiples of 16", (rng.integers(0, 16, (B, 1080, 1920, 3)) * 16).astype(np.uint8))
