#!/usr/bin/env python3
"""persistent kernel (stage 3 deferred into the next tile's piece copies) against the per-tile kernels: bytes and time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, _lib
import bench
eng = L.LerfEngine.shipped("lerf-g")
for (n, H, W, kind) in ((8, 1080, 1920, "noise"), (8, 1080, 1920, "natural"), (2, 2160, 3840, "noise"), (3, 517, 1003, "noise"), (16, 256, 256, "noise")):
    x = torch.from_numpy(bench.synth_frames(kind, n, 5, H, W)).cuda()
    geo = eng.sr_geometry((H, W), 2)
    assert geo.flags & _lib.GEO_X2_TABLES
    gold = geo.with_flags(_lib.GEO_NO_PERSIST)
    a = ops.sr_fused_u8(x, eng.luts, gold, "gauss", 10.0)
    b = ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0)
    torch.cuda.synchronize()
    diff = (a != b)
    print("%d x %dx%d %s: %d bytes differ of %d" % (n, H, W, kind, int(diff.sum()), a.numel()), end="")
    if diff.any():
        idx = torch.nonzero(diff)
        print("  first:", idx[:5].tolist(), "rows", int(idx[:, 1].min()), int(idx[:, 1].max()), "cols", int(idx[:, 2].min()), int(idx[:, 2].max()), end="")
    ts = []
    for g in (gold, geo):
        for _ in range(3): ops.sr_fused_u8(x, eng.luts, g, "gauss", 10.0, out=a)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): ops.sr_fused_u8(x, eng.luts, g, "gauss", 10.0, out=a)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 20)
    print("   per-tile %.3f ms, persistent %.3f ms (%.1f -> %.1f Gpix/s)" % (ts[0] * 1e3, ts[1] * 1e3, n * 4 * H * W / ts[0] / 1e9, n * 4 * H * W / ts[1] / 1e9))
