#!/usr/bin/env python3
"""Per-phase cycle breakdown of the tile-fused kernel (diagnostic build with
in-kernel s_memtime stamps; shares only the source with the shipped library).

    make -C lerf-pytorch_amd/csrc stamps && python tools/stamps.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LERF_HIP_LIB", os.path.join(ROOT, "lerf-pytorch_amd", "liblerf_hip_stamps.so"))

import numpy as np
import torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench

H, W = 1080, 1920

NAMES = ["in+lut0", "s1:s", "copy", "s1:c", "copy", "s1:t", "bin+slots", "s2 copies", "s2 lookups", "(s2 total)",
         "finalise", "geometry", "stage3"]


def main():
    model = os.environ.get("LERF_STAMPS_MODEL", "lerf-g")            # lerf-l: stage 2 + packing = stamps 6 -> 11 ("s2" rows are not set)
    scale = tuple(float(v) for v in os.environ.get("LERF_STAMPS_SCALE", "2").split(","))
    eng = L.LerfEngine.shipped(model)
    geo = eng.sr_geometry((H, W), scale if len(scale) > 1 else scale[0])
    for kind in (sys.argv[1:] or ("noise", "natural", "constant")):
        if kind == "constant":
            x = torch.full((1, H, W, 3), 128, dtype=torch.uint8, device="cuda")
        else:
            x = torch.from_numpy(bench.synth_frames(kind, 1, 7, H, W)).cuda()
        tiles = ((H + 63) // 64) * ((W + 63) // 64)
        ws = torch.zeros(max(tiles * 16 * 8, 4 * H * W * 3), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, workspace=ws)
        torch.cuda.synchronize()
        st = ws[:tiles * 16 * 8].view(torch.int64).reshape(tiles, 16).cpu().numpy().astype(np.float64)
        d = {}
        d["in+lut0"] = st[:, 1] - st[:, 0]
        d["s1:s"] = st[:, 2] - st[:, 1]
        d["copy1"] = st[:, 3] - st[:, 2]
        d["s1:c"] = st[:, 4] - st[:, 3]
        d["copy2"] = st[:, 5] - st[:, 4]
        d["s1:t"] = st[:, 6] - st[:, 5]
        d["bin+slots"] = st[:, 7] - st[:, 6]
        d["s2 copies"] = st[:, 8]
        d["s2 lookups"] = st[:, 9]
        d["s2 total"] = st[:, 10] - st[:, 7]
        d["finalise+geo"] = st[:, 11] - st[:, 10]
        d["stage 2 .. geometry (6 -> 11)"] = st[:, 11] - st[:, 6]
        d["stage3"] = st[:, 12] - st[:, 11]
        blkt = st[:, 15] > 0                                # tiles that took the block tasks of stage 3
        if blkt.any():
            d["s3 tasks (block tiles)"] = np.where(blkt, st[:, 15] - st[:, 11], np.nan)
            d["s3 ties+flush (block tiles)"] = np.where(blkt, st[:, 12] - st[:, 15], np.nan)
            d["stage3 (other tiles)"] = np.where(~blkt, st[:, 12] - st[:, 11], np.nan)
        d["TOTAL"] = st[:, 12] - st[:, 0]
        rt = st[:, 14] - st[:, 13]                       # s_memrealtime ticks (100 MHz) over the same interval as TOTAL
        mhz = d["TOTAL"] / np.maximum(rt, 1) * 100.0
        wall = (st[:, 14].max() - st[:, 13].min()) / 100.0          # us, first tile start -> last tile end
        print("== %s: cycles per tile (mean over %d tiles; s_memtime ticks, stamped on wave 0).  Shader clock during the tiles: "
              "s_memtime / s_memrealtime x 100 MHz = %.0f MHz mean (%.0f .. %.0f); launch = %.1f us by the 100 MHz counter"
              % (kind, tiles, mhz.mean(), mhz.min(), mhz.max(), wall))
        tot = d["TOTAL"].mean()
        for k, v in d.items():
            print("  %-28s %10.0f  %5.1f%%" % (k, np.nanmean(v), 100 * np.nanmean(v) / tot))


if __name__ == "__main__":
    main()
