#!/usr/bin/env python3
"""Thread scaling of the CPU port of the oracle (oracle/lerf_oracle.c) on this host: LeRF-G x2, one 1080p frame per call,
caller-owned scratch.  Prints Mpix/s per thread count, the cgroup CPU quota and the affinity mask (a container may see 256
processors and be allowed a fraction of them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from oracle import c_oracle, lerf_oracle as O

luts = O.load_luts(os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", "lerf-g"))
img = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
print("os.cpu_count()", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
base = None
for thr in [1, 8, 16, 32, 48, 64, 96, 128, 192, 256]:
    if thr > os.cpu_count():
        break
    c_oracle.set_threads(thr)
    out = c_oracle.sr_u8(img, luts, 2, 2)
    n, t0 = 0, time.perf_counter()
    while True:
        c_oracle.sr_u8(img, luts, 2, 2, out=out)
        n += 1
        if time.perf_counter() - t0 > (6 if thr == 1 else 3):
            break
    dt = (time.perf_counter() - t0) / n
    v = out.shape[0] * out.shape[1] / dt / 1e6
    base = base or v
    print("%4d threads: %8.2f Mpix/s  (x%.1f of one thread, %.0f %% per-thread efficiency)" % (thr, v, v / base, 100 * v / base / thr))
