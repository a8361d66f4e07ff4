#!/usr/bin/env python3
"""timing only: headline step (8 x 1080p, noise) through whatever kernel the library build takes"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, _lib
import bench
eng = L.LerfEngine.shipped("lerf-g")
x = torch.from_numpy(bench.synth_frames("noise", 8, 5, 1080, 1920)).cuda()
geo = eng.sr_geometry((1080, 1920), 2)
a = ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0)
for g, nm in ((geo.with_flags(_lib.GEO_NO_PERSIST), "per-tile kernel"), (geo, "persistent kernel")):
    for _ in range(3): ops.sr_fused_u8(x, eng.luts, g, "gauss", 10.0, out=a)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): ops.sr_fused_u8(x, eng.luts, g, "gauss", 10.0, out=a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
    print("%s %s: %.3f ms per step" % (os.path.basename(_lib.LIB_PATH), nm, dt * 1e3))
