#!/usr/bin/env python3
"""stamped sr_fused_kernel totals per tile column / row for one rank's block (region of interest, two launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
# (round 4: selected the stamped build through LERF_HIP_LIB; today: lerf_pytorch_amd._lib.use_library(path))
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, dist as ldist
eng = L.LerfEngine.shipped("lerf-g")
geo = eng.sr_geometry((2160, 3840), 2)
x = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (1, 2160, 3840, 3), dtype=np.uint8)).cuda()
np.set_printoptions(linewidth=250)
for r in [int(a) for a in sys.argv[1:]] or [0, 1]:
    plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, geo.host["left_r"], geo.host["left_c"])
    ext = x[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous()
    lg = ldist.block_geometry(geo, plan)
    lh, lw = plan.local_hw
    tiles = 17 * 15
    ws = torch.zeros(max(int(L._lib.lib().lerf_sr_fused_workspace_bytes(lh, lw, 3, 1)), ((2 * tiles + tiles // 4 + 64) * 128 + 255 & ~255) + lh * lw * 3 + 64), dtype=torch.uint8, device="cuda")
    for _ in range(2): ops.sr_fused_u8(ext, eng.luts, lg, "gauss", 10.0, workspace=ws)
    torch.cuda.synchronize()
    st = ws[:tiles * 128].view(torch.int64).reshape(17, 15, 16).cpu().numpy().astype(np.float64)
    tot = st[:, :, 12] - st[:, :, 0]
    print("rank %d roi %s: sr_fused_kernel cycles per tile (k), rows = tile rows" % (r, plan.roi,))
    print((tot / 1e3).round(0).astype(int))
    for nm, a, b in (("load", 0, 6), ("bin", 6, 7), ("s2", 7, 10), ("fin+geo", 10, 11), ("s3", 11, 12)):
        v = st[:, :, b] - st[:, :, a]
        print("  %-8s column means (k):" % nm, (v.mean(axis=0) / 1e3).round(1))
