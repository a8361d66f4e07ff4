#!/usr/bin/env python3
"""Hybrid host-to-host pipeline: the kernel reads the pinned input itself, writes device memory; a copy engine takes the result down."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, _lib
eng = L.LerfEngine.shipped("lerf-g")
B, H, W = 8, 1080, 1920
geo = eng.sr_geometry((H, W), 2); oH, oW = geo.out_hw
frame = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
nb = int(_lib.lib().lerf_sr_fused_workspace_bytes(H, W, 3, B))
def T(f, n=8):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
h_in = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory(); h_in.numpy()[:] = frame
d_in = h_in.cuda(); d_out = torch.empty((B, oH, oW, 3), dtype=torch.uint8, device="cuda"); ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
h_out = torch.empty((B, oH, oW, 3), dtype=torch.uint8).pin_memory()
print("SR device -> device: %.3f ms" % T(lambda: ops.sr_fused_u8(d_in, eng.luts, geo, eng.kind, eng.max_sigma, out=d_out, workspace=ws)))
print("SR pinned host -> device: %.3f ms" % T(lambda: ops.sr_fused_u8(h_in, eng.luts, geo, eng.kind, eng.max_sigma, out=d_out, workspace=ws)))
print("SR device -> pinned host: %.3f ms" % T(lambda: ops.sr_fused_u8(d_in, eng.luts, geo, eng.kind, eng.max_sigma, out=h_out, workspace=ws)))
for depth in (2, 3):
    comp, down = torch.cuda.Stream(), torch.cuda.Stream()
    slots = [dict(h_in=torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory(), h_out=torch.empty((B, oH, oW, 3), dtype=torch.uint8).pin_memory(),
                  d_out=torch.empty((B, oH, oW, 3), dtype=torch.uint8, device="cuda"), ws=torch.empty(nb, dtype=torch.uint8, device="cuda"),
                  sr=torch.cuda.Event(), done=torch.cuda.Event()) for _ in range(depth)]
    for s in slots: s["h_in"].numpy()[:] = frame
    def submit(i):
        s = slots[i]
        with torch.cuda.stream(comp):
            ops.sr_fused_u8(s["h_in"], eng.luts, geo, eng.kind, eng.max_sigma, out=s["d_out"], workspace=s["ws"]); s["sr"].record(comp)
        with torch.cuda.stream(down):
            down.wait_event(s["sr"]); s["h_out"].copy_(s["d_out"], non_blocking=True); s["done"].record(down)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); pend = []; stamps = []; nxt = 0
        for _ in range(12):
            if len(pend) == depth:
                slots[pend.pop(0)]["done"].synchronize(); stamps.append(time.perf_counter() - t0)
            submit(nxt); pend.append(nxt); nxt = (nxt + 1) % depth
        while pend:
            slots[pend.pop(0)]["done"].synchronize(); stamps.append(time.perf_counter() - t0)
        print("hybrid depth", depth, "total %.2f ms = %.3f ms per frame:" % (stamps[-1] * 1e3, stamps[-1] * 1e3 / (12 * B)), " ".join("%.1f" % (s * 1e3) for s in stamps))
    want = eng.sr(torch.from_numpy(frame).cuda(), 2).cpu().numpy()
    print("   bytes equal:", all(np.array_equal(s["h_out"].numpy()[b], want) for s in slots for b in (0, B - 1)))
