#!/usr/bin/env python3
"""cProfile of the lazy leg of the unchanged call sites (host image in), sorted by own time; plus a device-side kernel time sum."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import callsite_driver as cd
from oracle import lerf_oracle as O
img = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3)).astype(np.float32)
luts = cd.float_luts(O.load_luts(os.path.join(ROOT, "lerf-pytorch_amd/assets/models/lerf-g")))
interp, pads, resizer = cd.mirror_api()
for _ in range(2):
    out = np.asarray(cd.worker_sr(interp, pads, resizer, luts, img, (2.0, 2.0)))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
out = np.asarray(cd.worker_sr(interp, pads, resizer, luts, img, (2.0, 2.0)))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
# device time of the same worker: events around it with the host far ahead is not possible (sync copies); use the torch profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    out = np.asarray(cd.worker_sr(interp, pads, resizer, luts, img, (2.0, 2.0)))
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
