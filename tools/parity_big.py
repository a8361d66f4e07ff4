#!/usr/bin/env python3
"""Byte-level parity of the fused kernel against the float64 C oracle on large frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import lerf_pytorch_amd as L
from oracle import c_oracle, lerf_oracle as O
import bench
tot = bad = 0
for model, linear in (("lerf-g", False), ("lerf-l", True)):
    eng = L.LerfEngine.shipped(model)
    luts = O.load_luts(os.path.join(ROOT, "lerf-pytorch_amd/assets/models", model), linear=linear)
    for kind, seed, scale in (("noise", 11, 2), ("noise", 12, 2), ("natural", 13, 2), ("noise", 14, 3), ("natural", 15, 1.5)):
        img = bench.synth_frames(kind, 1, seed, 1080, 1920)[0]
        if scale != 2: img = img[:540, :960]
        out = eng.sr(img, scale)
        ref = c_oracle.sr_u8(img, luts, scale, scale, linear=linear)
        d = np.abs(out.astype(int) - ref.astype(int))
        tot += d.size; bad += int((d != 0).sum())
        print(model, kind, seed, "x%s" % scale, out.shape, "max diff", int(d.max()), "mismatched bytes", int((d != 0).sum()), "of", d.size)
print("TOTAL mismatched %d of %d bytes" % (bad, tot))
