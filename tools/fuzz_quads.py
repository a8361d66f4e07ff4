#!/usr/bin/env python3
"""Randomised sizes at the integer x2 scale with the 4 x 4 support: the quad tasks of stage 3 (round 5: 2 x 2 neighbouring blocks per
task on interior tiles; lerf_fused_impl.h run_quads) vs the C port of the oracle, bytes.  Sizes around the tile grid of all three
tile heights (64 / 32 / 16 rows), batches of 1..3 frames; S = 2 rides along (single blocks).   usage: fuzz_quads.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from oracle import c_oracle, lerf_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
luts = lerf_oracle.load_luts(os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", "lerf-g"))
engs = {2: L.LerfEngine.shipped("lerf-g", support=2), 4: L.LerfEngine.shipped("lerf-g", support=4)}
bad = tot = 0
for i in range(n):
    S = 4 if rng.random() < 0.8 else 2
    H, W = int(rng.integers(40, 700)), int(rng.integers(40, 900))
    if rng.random() < 0.4:
        H, W = int(rng.choice([64, 128, 192, 200, 256, 320])) + int(rng.integers(-2, 3)), int(rng.choice([64, 128, 192, 256, 448])) + int(rng.integers(-2, 3))
    nb = int(rng.integers(1, 4))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8) if rng.random() < 0.7 else \
        np.clip(np.add.outer(np.arange(H) * 3, np.arange(W) * 2)[..., None] + rng.integers(0, 9, (H, W, 3)), 0, 255).astype(np.uint8)
    x = torch.from_numpy(np.stack([img] * nb)).cuda()
    out = engs[S].sr(x, (2.0, 2.0)).cpu().numpy()
    ref = c_oracle.sr_u8(img, luts, 2.0, 2.0, S=S)
    d = out != ref[None]
    tot += d.size
    if d.any():
        bad += int(d.sum())
        print("MISMATCH S=%d %dx%d x%d frames: %d bytes" % (S, H, W, nb, int(d.sum())))
print("fuzz_quads: %d cases, %d bytes compared, %d mismatching" % (n, tot, bad))
sys.exit(1 if bad else 0)
