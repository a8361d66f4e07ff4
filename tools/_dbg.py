import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench
eng = L.LerfEngine.shipped("lerf-g")
img = bench.synth_frames("noise", 1, 77, 540, 960)
x = torch.from_numpy(img).cuda()
geo0 = eng.sr_geometry((540, 960), 2)
outs = []
for i in range(4):
    outs.append(ops.sr_fused_u8(x, eng.luts, geo0, eng.kind, eng.max_sigma).cpu().numpy()[0])
print("same geometry, 4 calls: diffs vs first", [int((o != outs[0]).sum()) for o in outs])
g = geo0.with_tie_queue_cap(100)
outs2 = [ops.sr_fused_u8(x, eng.luts, g, eng.kind, eng.max_sigma).cpu().numpy()[0] for i in range(3)]
print("cap 100, 3 calls: diffs vs default", [int((o != outs[0]).sum()) for o in outs2])
print("geo structs:", geo0.struct.tie_queue_cap, g.struct.tie_queue_cap)
import ctypes
for f, _ in geo0.struct._fields_:
    va, vb = getattr(geo0.struct, f), getattr(g.struct, f)
    if va != vb: print("  field", f, va, vb)
