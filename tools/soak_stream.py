#!/usr/bin/env python3
"""Soak of stream.StreamingSR (default transport): N batches of changing content through the reused slots, every output batch
checked against the synchronous engine's bytes for its input; prints the slowest batch intervals.   usage: soak_stream.py [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd.stream import StreamingSR
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 4, 540, 960
rng = np.random.default_rng(7)
pool = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(5)]
want = [eng.sr(torch.from_numpy(p).cuda(), 2).cpu().numpy() for p in pool]
for transport in ("dma", "zero_copy"):
    st = StreamingSR(eng, (H, W), 2, frames_per_batch=B, transport=transport)
    bad = 0; t_prev = time.perf_counter(); gaps = []
    for k, out in enumerate(st.run(pool[i % 5] for i in range(n))):
        t = time.perf_counter(); gaps.append(t - t_prev); t_prev = t
        if k % 7 == 0 or k > n - 10:
            bad += int(not np.array_equal(out, want[k % 5]))
    g = np.array(gaps[5:]) * 1e3
    print("%s: %d batches, checked %d, mismatching %d; interval median %.3f ms, p99 %.3f, max %.3f" % (transport, n, len(range(0, n, 7)), bad, np.median(g), np.percentile(g, 99), g.max()))
