"""How many uint8 outputs of the production resampler differ from the float64 oracle on ADVERSARIAL hyper-parameter maps, for a
given build of the library (the tie-guard epsilon is a build-time constant, LERF_TIE_EPS; tools/build_variant_all.sh):
   sat 0  uniform random bytes                  sat 1  sigma_x = sigma_y = 255 over the top third (k0 random)
   sat 2  all three parameters 255 over a third  sat 3  8 % of the bytes 255, 5 % 0, at random positions
usage: probe_tie_eps.py [path of a variant liblerf_hip.so]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lerf_pytorch_amd import ops, _lib
if len(sys.argv) > 1:
    _lib.use_library(sys.argv[1])
from oracle import lerf_oracle as O
tot = {}
for sat in (0, 1, 2, 3):
    nd = nb = 0
    worst = 0
    for (sh, sw) in [(2, 2), (3, 3), (4, 4), (1.5, 2.0), (2.7, 1.3)]:
        for seed in range(3):
            rng = np.random.default_rng(seed * 100 + int(sh * 10 + sw))
            H, W = int(rng.integers(20, 70)), int(rng.integers(20, 90))
            feat = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            hq = rng.integers(0, 256, (H, W, 3, 3), dtype=np.uint8)
            if sat == 1:
                hq[: H // 3, :, :, 1:] = 255
            if sat == 2:
                hq[: H // 3] = 255
            if sat == 3:
                hq[rng.random(hq.shape) < 0.08] = 255
                hq[rng.random(hq.shape) < 0.05] = 0
            geo = ops.SrGeometry((H, W), [sh, sw], None, 2)
            ref = O.to_u8(O.resize_u8(feat, hq, sh, sw, 2, 10.0, "gauss"))
            out = ops.resize_hwc_u8(torch.from_numpy(feat).cuda(), torch.from_numpy(hq).cuda(), geo, "gauss", 10.0, out="u8").cpu().numpy()
            d = np.abs(out.astype(int) - ref.astype(int))
            nd += int((d != 0).sum()); nb += d.size; worst = max(worst, int(d.max()))
    print("sat %d: %d of %d bytes differ (worst %d)" % (sat, nd, nb, worst))
