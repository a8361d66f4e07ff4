#!/usr/bin/env python3
"""Randomised homographies: LerfEngine.warp (fused stages + packed warp kernel) vs the numpy oracle (checker);
uint8 outputs must agree within 1 LSB on the valid region and the masks exactly; the tile-fused warp (lerf_warp_fused_u8, opt-in)
must equal the packed path byte for byte wherever it takes the call.   usage: fuzz_warps.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import lerf_pytorch_amd as L
from oracle import lerf_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
luts = {m: O.load_luts(os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", m), linear=(m == "lerf-l")) for m in ("lerf-g", "lerf-l")}
engs = {m: L.LerfEngine.shipped(m) for m in luts}
fengs = {m: L.LerfEngine.shipped(m) for m in luts}
for e in fengs.values():
    e.fused_warp = True
fused_bad = fused_n = 0
worst = 0
flips = tot = 0
for i in range(n):
    model = "lerf-g" if rng.random() < 0.6 else "lerf-l"
    H, W = int(rng.integers(12, 90)), int(rng.integers(12, 110))
    z = rng.uniform(0.8, 4.0)
    th = rng.uniform(-0.3, 0.3)
    M = np.array([[z * np.cos(th), -z * np.sin(th) * rng.uniform(0.5, 1.5), rng.uniform(-10, 30)],
                  [z * np.sin(th), z * np.cos(th) * rng.uniform(0.7, 1.3), rng.uniform(-10, 30)],
                  [rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 1.0]])
    oh, ow = int(rng.integers(10, 200)), int(rng.integers(10, 240))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    out, mask = engs[model].warp(img, M, (oh, ow))
    fout, fmask = fengs[model].warp(img, M, (oh, ow))
    fused_n += 1
    if not (np.array_equal(fout, out) and np.array_equal(fmask, mask)):
        fused_bad += 1
        print("FUSED DIFF", model, (H, W), (oh, ow), int((fout != out).sum()), M.tolist())
    ref = O.warp_pipeline(img, luts[model], M, (oh, ow), linear=(model == "lerf-l"))
    rmask = O.warp_mask((H, W), M, (oh, ow))
    if not np.array_equal(mask, rmask):
        print("MASK DIFF", model, (H, W), (oh, ow), int((mask != rmask).sum()))
    d = np.abs(out.astype(int) - ref.astype(int)) * rmask
    worst = max(worst, int(d.max()))
    flips += int((d != 0).sum())
    tot += int(rmask.sum())
    if d.max() > 1:
        print("DIFF", model, (H, W), (oh, ow), d.max(), M.tolist())
print("tile-fused warp == packed path: %d cases, %d differing" % (fused_n, fused_bad))
print("cases %d, valid bytes %d, 1-LSB flips %d (%.2e), worst %d" % (n, tot, flips, flips / max(tot, 1), worst))
