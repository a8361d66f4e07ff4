#!/usr/bin/env python3
"""Where does a stage-1 call of the unchanged call sites spend its time? (host numpy, host -> pinned, H2D, device ops)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
from lerf_pytorch_amd import lazy, ops, _lib
from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster
print("torch threads", torch.get_num_threads(), "cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
img = np.random.default_rng(0).integers(0, 256, (1080, 1920, 3)).astype(np.float32)
def T(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
pad = lambda: np.pad(np.rot90(img, 1), ((0, 3), (0, 3), (0, 0)), mode="edge")
print("caller: rot90 + pad(edge): %.2f ms" % T(pad))
p = pad()
pin = torch.empty(p.nbytes, dtype=torch.uint8).pin_memory()
pv = pin[:p.nbytes].view(torch.float32).reshape(p.shape)
print("torch copy_ pageable -> pinned: %.2f ms" % T(lambda: pv.copy_(torch.from_numpy(p))))
pn = pv.numpy()
print("np.copyto pageable -> pinned:   %.2f ms" % T(lambda: np.copyto(pn, p)))
print("pinned -> device (25 MB):       %.2f ms" % T(lambda: pv.to("cuda", non_blocking=True)))
print("pageable .cuda() (25 MB):       %.2f ms" % T(lambda: torch.from_numpy(p).cuda()))
u8 = lambda: p.astype(np.uint8)
print("host float32 -> uint8:          %.2f ms" % T(u8))
q = u8()
print("pageable .cuda() (6 MB u8):     %.2f ms" % T(lambda: torch.from_numpy(q).cuda()))
print("lazy.upload(p):                 %.2f ms" % T(lambda: lazy.upload(p)))
d = lazy.upload(p).permute(2, 0, 1)
w = np.load(os.path.join(ROOT, "lerf-pytorch_amd/assets/models/lerf-g/LUTft_s1_cr0.npy")).reshape(-1, 1).astype(np.float32)
chw = p.transpose((2, 0, 1))
print("FourSimplexInterpFaster(numpy in, lazy out):  %.2f ms" % T(lambda: FourSimplexInterpFaster(w, chw, 1920, 1080, 4, 3, upscale=1, mode="c", oC=1)))
da = lazy.DeviceArray(d)
print("FourSimplexInterpFaster(device in, lazy out): %.2f ms" % T(lambda: FourSimplexInterpFaster(w, da, 1920, 1080, 4, 3, upscale=1, mode="c", oC=1)))
for nt in (1, 4, 16):
    torch.set_num_threads(nt)
    print("threads %d: torch copy_ pageable -> pinned: %.2f ms" % (nt, T(lambda: pv.copy_(torch.from_numpy(p)))))

# ---- the real flow of stage 1: fresh rot90 + pad per call, per-step times
torch.set_num_threads(128)
import callsite_driver as cd
from oracle import lerf_oracle as O
luts = cd.float_luts(O.load_luts(os.path.join(ROOT, "lerf-pytorch_amd/assets/models/lerf-g")))
interp, pads, resizer = cd.mirror_api()
for rep in range(2):
    tt = {"caller": 0.0, "interp": 0.0, "sync": 0.0}
    total = 0
    for m in "sct":
        for r in range(4):
            t0 = time.perf_counter()
            rot = np.rot90(img, r); h, w, _ = rot.shape
            chw = np.pad(rot, ((0, pads[m]), (0, pads[m]), (0, 0)), mode="edge").transpose((2, 0, 1))
            t1 = time.perf_counter()
            total += interp(luts["s1_%sr0" % m], chw, h, w, 4, 4 - r, upscale=1, mode=m, oC=1)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            tt["caller"] += t1 - t0; tt["interp"] += t2 - t1; tt["sync"] += t3 - t2
    print("stage 1, 12 calls: caller numpy %.1f ms, interp (host side) %.1f ms, waiting for the device %.1f ms" % (tt["caller"] * 1e3, tt["interp"] * 1e3, tt["sync"] * 1e3))
t0 = time.perf_counter()
out = np.asarray(cd.worker_sr(interp, pads, resizer, luts, img, (2.0, 2.0)))
print("whole worker_sr: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
out = np.asarray(cd.worker_sr(interp, pads, resizer, luts, img, (2.0, 2.0)))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
