#!/usr/bin/env python3
"""Why does eng.sr(torch.randint frames) run faster than the bench's noise step?  Same box, same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
import bench
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 8, 1080, 1920
def T(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
xr = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
xn = torch.from_numpy(np.stack(bench.synth_frames("noise", B, 1000, H, W))).cuda()
xg = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
x1 = xg[:1].expand(B, H, W, 3).contiguous()
for name, x in (("torch.randint", xr), ("bench noise", xn), ("numpy integers", xg), ("one numpy frame x 8", x1), ("torch.randint again", xr)):
    hist = torch.bincount(x.flatten().to(torch.int64), minlength=256).float()
    print("%-22s %.3f ms per 8 frames; byte histogram min/max %d/%d, mean %.2f" % (name, T(lambda: eng.sr(x, 2)), hist.min().item(), hist.max().item(), x.float().mean().item()))
