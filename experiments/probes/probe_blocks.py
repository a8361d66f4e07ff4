#!/usr/bin/env python3
"""One rank's batched block launch pair (config 5, 2 x 4 grid), for rocprofv3 --kernel-trace: tools/probe_blocks.py <rank>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, dist as ldist
r = int(sys.argv[1])
eng = L.LerfEngine.shipped("lerf-g")
geo = eng.sr_geometry((2160, 3840), 2)
x8 = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (8, 2160, 3840, 3), dtype=np.uint8)).cuda()
plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, geo.host["left_r"], geo.host["left_c"])
ext8 = x8[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous()
lg = ldist.block_geometry(geo, plan)
print("rank", r, "local", plan.local_hw, "roi", plan.roi)
o = ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0)
for _ in range(10): ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=o)
torch.cuda.synchronize()
