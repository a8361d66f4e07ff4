#!/usr/bin/env python3
"""Randomised shape / scale sweep: fused uint8 SR path vs the C port of the oracle (checker).  Not part of the
test suite (takes a minute); prints every case that differs.    usage: fuzz_shapes.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from oracle import c_oracle, lerf_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
luts = {m: lerf_oracle.load_luts(os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", m), linear=(m == "lerf-l")) for m in ("lerf-g", "lerf-l")}
engs = {("lerf-g", 2): L.LerfEngine.shipped("lerf-g", support=2), ("lerf-g", 4): L.LerfEngine.shipped("lerf-g", support=4),
        ("lerf-l", 2): L.LerfEngine.shipped("lerf-l")}
bad = tot = 0
for i in range(n):
    model, S = [("lerf-g", 2), ("lerf-g", 2), ("lerf-g", 4), ("lerf-l", 2)][rng.integers(0, 4)]
    H, W = int(rng.integers(1, 260)), int(rng.integers(1, 330))
    if rng.random() < 0.3:
        W = (W + 3) // 4 * 4                       # aligned-dword input path
    sh, sw = [float(rng.choice([1.0, 1.3, 1.5, 2.0, 2.4, 3.0, 4.0])) for _ in range(2)]
    kind = rng.integers(0, 3)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if kind == 0 else \
        (np.full((H, W, 3), rng.integers(0, 256), np.uint8) if kind == 1 else
         np.clip(np.add.outer(np.arange(H) * 3, np.arange(W) * 2)[..., None] + rng.integers(0, 9, (H, W, 3)), 0, 255).astype(np.uint8))
    nb = int(rng.integers(1, 4))
    x = torch.from_numpy(np.stack([img] * nb)).cuda()
    out = engs[(model, S)].sr(x, (sh, sw)).cpu().numpy()
    ref = c_oracle.sr_u8(img, luts[model], sh, sw, S=S, linear=(model == "lerf-l"))
    d = np.abs(out.astype(int) - ref[None].astype(int))
    tot += d.size
    if d.max() > 0:
        bad += int((d != 0).sum())
        print("DIFF", model, S, (H, W), (sh, sw), "frames", nb, "max", d.max(), "count", int((d != 0).sum()))
print("cases %d, bytes %d, mismatched %d" % (n, tot, bad))
