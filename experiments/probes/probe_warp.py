#!/usr/bin/env python3
"""config 4 step (8 x 1080p -> 4K, isc matrix) kernel by kernel: events around the stages launch pair and the warp launch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench
eng = L.LerfEngine.shipped("lerf-g")
x = torch.from_numpy(bench.synth_frames("natural", 8, 1000, 1080, 1920)).cuda()
geo = ops.WarpGeometry((1080, 1920), np.array(bench.M_ISC), (2160, 3840), 2)
out = torch.empty((8, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
ws = torch.empty(L._lib.lib().lerf_sr_fused_workspace_bytes(1080, 1920, 3, 8), dtype=torch.uint8, device="cuda")
packed = ops.stages_packed(x, eng.luts, workspace=ws)
def T(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
ts = T(lambda: ops.stages_packed(x, eng.luts, workspace=ws))
tw = T(lambda: ops.warp_packed(packed, geo, "gauss", 10.0, out=out))
print("%s: stages (s1 + EMIT) %.3f ms, packed warp %.3f ms, step %.3f ms = %.1f Gpix/s" % (os.path.basename(_lib.LIB_PATH), ts, tw, ts + tw, 8 * 2160 * 3840 / (ts + tw) / 1e6))
