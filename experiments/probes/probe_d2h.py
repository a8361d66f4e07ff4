#!/usr/bin/env python3
"""D2H / H2D copy rates to pinned memory, alone and beside a running SR launch (can a copy-engine pipeline beat the zero-copy kernel?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 8, 1080, 1920
x = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
y = eng.sr(x, 2)
hin = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
hout = torch.empty(tuple(y.shape), dtype=torch.uint8).pin_memory()
mb_out, mb_in = y.numel() / 1e6, x.numel() / 1e6
side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
def T(f, n=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
t = T(lambda: hout.copy_(y, non_blocking=True)); print("D2H %.0f MB default stream: %.3f ms = %.1f GB/s" % (mb_out, t * 1e3, mb_out / t / 1e3))
def on(s, f):
    with torch.cuda.stream(s): f()
t = T(lambda: on(side, lambda: hout.copy_(y, non_blocking=True))); print("D2H side stream: %.3f ms = %.1f GB/s" % (t * 1e3, mb_out / t / 1e3))
t = T(lambda: x.copy_(hin, non_blocking=True)); print("H2D %.0f MB default stream: %.3f ms = %.1f GB/s" % (mb_in, t * 1e3, mb_in / t / 1e3))
def both():
    on(side, lambda: hout.copy_(y, non_blocking=True)); on(side2, lambda: x.copy_(hin, non_blocking=True))
t = T(both); print("D2H + H2D on two side streams: %.3f ms" % (t * 1e3))
t = T(lambda: eng.sr(x, 2)); print("SR kernel alone (8 frames): %.3f ms" % (t * 1e3))
y2 = torch.empty_like(y); x2 = x.clone()
def overlapped():
    eng.sr(x2, 2)                                              # default stream
    on(side, lambda: hout.copy_(y, non_blocking=True))          # previous batch's result goes down meanwhile
    on(side2, lambda: x.copy_(hin, non_blocking=True))          # next batch's input comes up meanwhile
t = T(overlapped); print("SR kernel + D2H + H2D of neighbouring batches on three streams: %.3f ms per batch = %.3f ms per frame" % (t * 1e3, t * 1e3 / B))
def serial():
    x.copy_(hin, non_blocking=True); yy = eng.sr(x, 2); hout.copy_(yy, non_blocking=True)
t = T(serial); print("H2D, SR, D2H on one stream: %.3f ms per batch = %.3f ms per frame" % (t * 1e3, t * 1e3 / B))
from lerf_pytorch_amd.stream import StreamingSR
st = StreamingSR(eng, (H, W), 2, frames_per_batch=B, depth=2, transport="zero_copy")
def zc():
    st.result(st.submit())
t = T(zc); print("zero-copy kernel, one batch at a time: %.3f ms per batch = %.3f ms per frame" % (t * 1e3, t * 1e3 / B))

# ---- does a cross-stream event wait in front of the copy change how the copy runs beside a kernel?
xn = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
yb = [torch.empty_like(y) for _ in range(2)]
ev = torch.cuda.Event()
main = torch.cuda.current_stream()
def with_event_wait():
    for k in range(4):
        from lerf_pytorch_amd import ops
        yy = eng.sr(xn, 2)                                       # default stream
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            hout.copy_(yy, non_blocking=True)
t = T(with_event_wait, 5); print("4 x (SR on the default stream; D2H on a side stream behind an event wait): %.3f ms per batch" % (t * 1e3 / 4))
def host_driven():
    prev = None
    for k in range(4):
        yy = eng.sr(xn, 2)
        e = torch.cuda.Event(); e.record(main)
        if prev is not None:
            prev[0].synchronize()
            with torch.cuda.stream(side):
                hout.copy_(prev[1], non_blocking=True)
        prev = (e, yy)
    prev[0].synchronize()
    with torch.cuda.stream(side):
        hout.copy_(prev[1], non_blocking=True)
t = T(host_driven, 5); print("4 x (SR; the host waits for the previous SR, then issues its D2H on the side stream): %.3f ms per batch" % (t * 1e3 / 4))
t = T(lambda: eng.sr(xn, 2)); print("SR alone, noise: %.3f ms" % (t * 1e3))
