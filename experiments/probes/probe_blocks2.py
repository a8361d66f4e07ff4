#!/usr/bin/env python3
"""block launch pair time against the local row pitch (BlockPlan align) for an edge and a middle rank"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, dist as ldist
eng = L.LerfEngine.shipped("lerf-g")
geo = eng.sr_geometry((2160, 3840), 2)
x8 = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (8, 2160, 3840, 3), dtype=np.uint8)).cuda()
for r in (0, 1, 3, 5):
    for align in (4, 16, 64):
        plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, geo.host["left_r"], geo.host["left_c"], align=align)
        ext8 = x8[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous()
        lg = ldist.block_geometry(geo, plan)
        o = ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=o)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print("rank %d align %2d local %s roi %s: %.3f ms" % (r, align, plan.local_hw, plan.roi, dt * 1e3))
