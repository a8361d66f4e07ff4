import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from lerf_pytorch_amd import ops, _lib
if len(sys.argv) > 1:
    _lib.use_library(sys.argv[1])
from oracle import lerf_oracle as O
for sat in (0, 1, 2):
  for (sh, sw) in [(2,2),(3,3),(1.5,2.0)]:
    for C in (1,3):
        rng = np.random.default_rng(int(sh * 100 + sw * 10 + C))
        H, W = int(rng.integers(5, 70)), int(rng.integers(5, 90))
        feat = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        hq = rng.integers(0, 256, (H, W, C, 3), dtype=np.uint8)
        if sat == 1:
            hq[: H // 3, :, :, 1:] = 255
        if sat == 2:
            hq[: H // 3] = 255
            hq[H // 3: H // 2, :, :, 0] = 0
        geo = ops.SrGeometry((H, W), [sh, sw], None, 2)
        ref64 = O.resize_u8(feat, hq, sh, sw, 2, 10.0, "gauss")
        ref = O.to_u8(ref64)
        out = ops.resize_hwc_u8(torch.from_numpy(feat).cuda(), torch.from_numpy(hq).cuda(), geo, "gauss", 10.0, out="u8").cpu().numpy()
        o32 = ops.resize_hwc_u8(torch.from_numpy(feat).cuda(), torch.from_numpy(hq).cuda(), geo, "gauss", 10.0, out="f32").cpu().numpy()
        d = out != ref
        print("sat", sat, (sh, sw), C, (H, W), "diff", int(d.sum()), "of", d.size, "nan ref", int(np.isnan(ref64).sum()),
              "max |f32(f64 path) - ref64|", float(np.nanmax(np.abs(o32 - ref64))))
        if d.sum():
            ii = np.argwhere(d)[:4]
