#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (3: LeRF-L anisotropic SR, 4: LeRF-G homographic warp,
plus S=4 and the stage-only API) on one MI355X; inputs per SURVEY.md 8(d).  Prints one line per case."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

res = []
B = 8
for kind in ("noise", "natural"):
    x = torch.from_numpy(bench.synth_frames(kind, B, 5)).cuda()
    for model, S in (("lerf-g", 2), ("lerf-g", 4), ("lerf-l", 2)):
        eng = L.LerfEngine.shipped(model, support=S)
        for sc in ((2.0, 2.0), (1.5, 2.0), (3.0, 3.0)):
            if model == "lerf-g" and S == 4 and sc != (2.0, 2.0): continue
            geo = eng.sr_geometry((bench.H, bench.W), sc)
            out = torch.empty((B,) + geo.out_hw + (3,), dtype=torch.uint8, device="cuda")
            dt = timeit(lambda: ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, out=out))
            res.append(("SR %s S=%d x%.1f/%.1f %s" % (model, eng.support, sc[0], sc[1], kind), B * geo.out_hw[0] * geo.out_hw[1] / dt / 1e6, dt * 1e3 / B))
    eng = L.LerfEngine.shipped("lerf-g")
    dt = timeit(lambda: ops.stages_packed(x, eng.luts))
    res.append(("stages 1+2 only (packed) lerf-g %s [LR Mpix/s]" % kind, B * bench.H * bench.W / dt / 1e6, dt * 1e3 / B))
    Ms = {"isc-like": [[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]],
          "osc-like": [[4.1, 0.4, 30.0], [0.5, 3.8, 25.0], [8e-5, 1.2e-4, 1.0]]}
    for name, M in Ms.items():
        f = x[0]
        dt = timeit(lambda: eng.warp(f, np.array(M), (2160, 3840), return_mask=False), n=5)
        o, mask = eng.warp(f, np.array(M), (2160, 3840))
        res.append(("warp lerf-g %s %s (valid %.0f%%)" % (name, kind, 100 * float(mask.float().mean())), 2160 * 3840 / dt / 1e6, dt * 1e3))
# fixed-kernel baselines on the same 1080p -> 4K uint8 frames ("as fast as interpolation", README.md:45)
x = torch.from_numpy(bench.synth_frames("noise", 1, 5)).cuda()[0]
for kind, S in (("bilinear", 2), ("cubic", 4), ("lanczos3", 6)):
    geo = ops.SrGeometry((bench.H, bench.W), [2.0, 2.0], None, S)
    dt = timeit(lambda: ops.resize_hwc_u8(x, None, geo, kind, 1.0, out="u8"))
    res.append(("fixed-kernel %s S=%d x2.0/2.0 noise (direct kernel)" % (kind, S), geo.out_hw[0] * geo.out_hw[1] / dt / 1e6, dt * 1e3))
for name, mp, ms in res:
    print("%-58s %10.1f Mpix/s   %.3f ms/frame" % (name, mp, ms))
