#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-facing boundary (host buffer in, host buffer out), one 1080p frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
import bench
eng = L.LerfEngine.shipped("lerf-g")
img = bench.synth_frames("noise", 1, 3, 1080, 1920)[0]
for _ in range(3): eng.sr(img, 2)
torch.cuda.synchronize(); t = time.perf_counter(); n = 20
for _ in range(n): out = eng.sr(img, 2)
dt = (time.perf_counter() - t) / n
print("numpy->numpy LerfEngine.sr, 1920x1080->3840x2160, pageable host buffers: %.3f ms/frame = %.1f Mpix/s (H2D 6.2 MB + kernel + D2H 24.9 MB + sync)" % (dt * 1e3, out.shape[0] * out.shape[1] / dt / 1e6))
pin_in = torch.from_numpy(img).pin_memory(); pin_out = torch.empty((2160, 3840, 3), dtype=torch.uint8).pin_memory()
x = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n):
    x.copy_(pin_in, non_blocking=True); o = eng.sr(x, 2); pin_out.copy_(o, non_blocking=True); torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print("pinned host buffers, async copies on one stream: %.3f ms/frame = %.1f Mpix/s" % (dt * 1e3, 2160 * 3840 / dt / 1e6))

from lerf_pytorch_amd.stream import StreamingSR
for B in (1, 4, 8):
    st = StreamingSR(eng, (1080, 1920), 2, frames_per_batch=B, depth=2, transport="zero_copy")
    for k in range(st.depth): st.input(k)[:] = img                # the producer writes straight into the pinned inputs
    for _ in range(3): st.result(st.submit())
    nb = max(6, 48 // B)
    torch.cuda.synchronize(); t = time.perf_counter()
    pend = []
    for _ in range(nb):
        if len(pend) == st.depth: st.result(pend.pop(0))
        pend.append(st.submit())
    while pend: st.result(pend.pop(0))
    dt = (time.perf_counter() - t) / (nb * B)
    print("StreamingSR zero-copy (kernel reads / writes pinned host memory, %d frame(s) per launch): %.3f ms/frame = %.1f Mpix/s"
          % (B, dt * 1e3, 2160 * 3840 / dt / 1e6))
