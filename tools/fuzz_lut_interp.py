#!/usr/bin/env python3
"""Randomised single LUT passes (lerf_lut_interp_ex, ABI 7): the LDS-resident kernel -- tile forced to 32 / 64 rows or chosen by the
library, plane-form LUT or interleaved -- against the direct kernel on the same operands (which tests/test_gpu_lut_interp.py pins
to the oracle), every output type, rotation, sampling pattern, rotated offsets with clamped coordinates (what the lazy call sites
pass), operand layouts and the accumulate form.  Bit-exact or it prints the case.   usage: fuzz_lut_interp.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda")
luts = {oC: torch.from_numpy(rng.integers(-128, 128, (17 ** 4, oC), dtype=np.int8)).to(dev) for oC in (1, 3)}    # the reference's oC
planes = {oC: ops.lut_planes(l) for oC, l in luts.items()}
bad = tot = declined = 0
for i in range(n):
    oC = int(rng.choice([1, 3]))
    Cn = int(rng.choice([1, 3, 3, 4, 5]))                     # 5 channels: the LDS kernel declines, the library falls back
    big = rng.random() < 0.3
    h, w = (int(rng.integers(200, 700)), int(rng.integers(200, 900))) if big else (int(rng.integers(1, 140)), int(rng.integers(1, 300)))
    mode = "sctdy"[int(rng.integers(0, 5))]
    krot = int(rng.integers(0, 4))
    dy, dx = _lib.mode_offsets(mode, krot)
    # the frame is 0..3 rows / columns larger than the window: coordinates outside it clamp (both kernels' contract), so rotated
    # offsets (negative ones too) and short frames exercise the edge tiles
    Hp, Wp = h + int(rng.integers(0, 4)), w + int(rng.integers(0, 4))
    img8 = torch.from_numpy(rng.integers(0, 256, (Cn, Hp, Wp), dtype=np.uint8)).to(dev)
    layout = int(rng.integers(0, 4))
    if layout == 0: x = img8
    elif layout == 1: x = img8.to(torch.float32)
    elif layout == 2: x = img8.permute(1, 2, 0).contiguous().to(torch.float32).permute(2, 0, 1)        # HWC buffer seen as CHW
    else: x = img8.permute(1, 2, 0).contiguous().permute(2, 0, 1)                                      # uint8 HWC
    dt = [torch.float64, torch.float32, torch.int16][int(rng.integers(0, 3))]
    rot = int(rng.integers(0, 4))
    acc = rng.random() < 0.5
    kern = ["lds", "lds32", "lds64", None][int(rng.integers(0, 4))]
    pl = planes[oC] if (rng.random() < 0.5 and oC > 1) else None
    oh, ow = (h, w) if rot % 2 == 0 else (w, h)
    if acc:
        base = torch.from_numpy(rng.integers(-500, 500, (Cn * oC, oh, ow)).astype(np.int16)).to(dev).to(dt)
        a, b = base.clone(), base.clone()
    else:
        a = b = None
    try:
        want = ops.lut_interp(x, h, w, dy, dx, luts[oC], 4, rot=rot, out_dtype=dt, out=a, accumulate=acc, kernel="direct")
        got = ops.lut_interp(x, h, w, dy, dx, luts[oC], 4, rot=rot, out_dtype=dt, out=b, accumulate=acc, kernel=kern, planes=pl)
    except _lib.LerfError as e:                                  # a forced LDS kernel may refuse a shape (tiny frames): say so, go on
        if kern is None:
            raise
        declined += 1
        continue
    torch.cuda.synchronize()
    tot += want.numel() * want.element_size()
    if not torch.equal(want, got):
        bad += 1
        print("DIFF", dict(oC=oC, C=Cn, h=h, w=w, mode=mode, krot=krot, layout=layout, dt=str(dt), rot=rot, acc=acc, kern=kern, planar=pl is not None),
              int((want != got).sum()))
print("fuzz_lut_interp: %d cases (%d declined by a forced LDS kernel), %d bytes compared, %d mismatching cases" % (n, declined, tot, bad))
sys.exit(1 if bad else 0)
