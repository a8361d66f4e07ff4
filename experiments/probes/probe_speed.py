#!/usr/bin/env python3
"""Input dependence of the headline step (8 x 1080p -> 4K, LeRF-G): how much of it is LDS bank conflicts?  A constant frame turns every
LUT gather into a broadcast (all lanes of a wave read one address): what is left is VALU issue, the piece copies and stage 3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
import bench
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 8, 1080, 1920
def T(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
cases = [("uniform noise (bench)", torch.from_numpy(np.stack(bench.synth_frames("noise", B, 1000, H, W))).cuda()),
         ("natural-like (bench)", torch.from_numpy(np.stack(bench.synth_frames("natural", B, 1000, H, W))).cuda()),
         ("constant 0", torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda")),
         ("constant 137", torch.full((B, H, W, 3), 137, dtype=torch.uint8, device="cuda")),
         ("horizontal ramp", (torch.arange(W, device="cuda") % 256).to(torch.uint8).view(1, 1, W, 1).expand(B, H, W, 3).contiguous()),
         ("noise in 16 x 16 blocks", torch.from_numpy(np.kron(np.random.default_rng(3).integers(0, 256, (B, H // 8 // 2 + 1, W // 16, 3), dtype=np.uint8),
                                                            np.ones((1, 16, 16, 1), np.uint8))[:, :H, :W]).contiguous().cuda())]
for name, x in cases:
    ms = T(lambda: eng.sr(x, 2))
    print("%-26s %.3f ms per step = %.1f Gpix/s" % (name, ms, B * 4 * H * W / ms / 1e6))
