#!/usr/bin/env python3
"""StreamingSR transports: per-batch completion times of a 12-batch run (is the three-stream pipeline overlapping?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd.stream import StreamingSR
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 8, 1080, 1920
frame = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
for transport, depth in (("zero_copy", 2), ("dma", 2), ("dma", 3), ("dma", 4), ("dma", 3)):
    st = StreamingSR(eng, (H, W), 2, frames_per_batch=B, depth=depth, transport=transport)
    for k in range(st.depth): st.input(k)[:] = frame
    for _ in range(3): st.result(st.submit())
    torch.cuda.synchronize(); t0 = time.perf_counter(); pend = []; stamps = []
    for _ in range(12):
        if len(pend) == st.depth:
            st.result(pend.pop(0)); stamps.append(time.perf_counter() - t0)
        pend.append(st.submit())
    while pend:
        st.result(pend.pop(0)); stamps.append(time.perf_counter() - t0)
    print(transport, "depth", depth, "total %.2f ms = %.3f ms per frame; completion times (ms):" % (stamps[-1] * 1e3, stamps[-1] * 1e3 / (12 * B)), " ".join("%.1f" % (s * 1e3) for s in stamps))
    del st
