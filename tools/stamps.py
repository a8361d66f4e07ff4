#!/usr/bin/env python3
"""Per-phase cycle breakdown of the tile-fused kernels (diagnostic build with in-kernel s_memtime stamps; shares only the
source with the shipped library).

    make -C lerf-pytorch_amd/csrc stamps && python tools/stamps.py [noise natural constant] [--single]

Default: the TWO-launch path the headline runs (s1_kernel + sr_fused_kernel<FROM_FEAT>), one 1080p frame, with the tiles
split into classes (interior / frame edge / last tile row).  --single: the single-launch kernel (stage 1 per tile halo).
Env: LERF_STAMPS_MODEL (lerf-g), LERF_STAMPS_SCALE ("2" or "1.5,2"), LERF_STAMPS_SUPPORT (2).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import lerf_pytorch_amd as L
L._lib.use_library(os.environ.get("LERF_STAMPS_LIB") or os.path.join(ROOT, "lerf-pytorch_amd", "csrc", "build_stamps", "liblerf_hip_stamps.so"))
from lerf_pytorch_amd import ops
import bench

H, W = 1080, 1920


def classes(ty, tx):
    """tile class of every tile of the (ty x tx) grid: 0 interior, 1 first tile row, 3 first column, 4 last column, 2 last tile row"""
    c = np.zeros((ty, tx), np.int32)
    c[0, :] = 1
    c[:, 0] = 3
    c[:, -1] = 4
    c[-1, :] = 2
    return c.reshape(-1)


def report(kind, d, cls, mhz, wall, title):
    names = {0: "interior", 1: "top row", 3: "left column", 4: "right column", 2: "last row (56/64)"}
    print("== %s, %s: cycles per tile (s_memtime ticks, stamped on wave 0).  Shader clock during the tiles: %.0f MHz mean (%.0f .. %.0f); "
          "launch = %.1f us by the 100 MHz counter" % (kind, title, mhz.mean(), mhz.min(), mhz.max(), wall))
    sel = [("all %d tiles" % len(cls), np.ones(len(cls), bool))] + [("%s (%d)" % (names[k], (cls == k).sum()), cls == k) for k in (0, 1, 3, 4, 2)]
    print("  %-30s" % "" + "".join("%20s" % n for n, _ in sel))
    for k, v in d.items():
        print("  %-30s" % k + "".join("%20.0f" % (np.nanmean(v[m]) if m.any() and not np.isnan(v[m]).all() else float("nan")) for _, m in sel))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    single = "--single" in sys.argv
    model = os.environ.get("LERF_STAMPS_MODEL", "lerf-g")
    scale = tuple(float(v) for v in os.environ.get("LERF_STAMPS_SCALE", "2").split(","))
    S = int(os.environ.get("LERF_STAMPS_SUPPORT", "2"))
    eng = L.LerfEngine.shipped(model, support=S)
    geo = eng.sr_geometry((H, W), scale if len(scale) > 1 else scale[0])
    ty, tx = (H + 63) // 64, (W + 63) // 64
    nfr = int(os.environ.get("LERF_STAMPS_FRAMES", "1"))          # >= 2 frames of 1080p: the persistent kernel takes the launch
    tiles = ty * tx * nfr
    cls = np.tile(classes(ty, tx), nfr)
    for kind in (args or ("noise", "natural")):
        if kind == "constant":
            x = torch.full((nfr, H, W, 3), 128, dtype=torch.uint8, device="cuda")
        else:
            x = torch.from_numpy(bench.synth_frames(kind, nfr, 7, H, W)).cuda()
        stamp_bytes = ((2 * tiles + tiles // 4 + 64) * 128 + 255) & ~255
        feat_bytes = nfr * ((H * W * 3 + 15) // 16 * 16)
        need = int(L._lib.lib().lerf_sr_fused_workspace_bytes(H, W, 3, nfr))
        ws = torch.zeros(need + stamp_bytes + 4096, dtype=torch.uint8, device="cuda")
        g = geo.with_flags(L._lib.GEO_SINGLE_LAUNCH) if single else geo
        for _ in range(2):
            ops.sr_fused_u8(x, eng.luts, g, eng.kind, eng.max_sigma, workspace=ws)
        torch.cuda.synchronize()
        st = ws[:tiles * 128].view(torch.int64).reshape(tiles, 16).cpu().numpy().astype(np.float64)
        d = {}
        if single:
            d["in+lut0"] = st[:, 1] - st[:, 0]
            d["s1:s"] = st[:, 2] - st[:, 1]
            d["copy1"] = st[:, 3] - st[:, 2]
            d["s1:c"] = st[:, 4] - st[:, 3]
            d["copy2"] = st[:, 5] - st[:, 4]
            d["s1:t"] = st[:, 6] - st[:, 5]
        else:
            s1 = ws[tiles * 128:2 * tiles * 128].view(torch.int64).reshape(tiles, 16).cpu().numpy().astype(np.float64)
            d["s1_kernel: LUT phases"] = s1[:, 1] - s1[:, 0]
            d["s1_kernel: feat store"] = s1[:, 2] - s1[:, 1]
            d["s1_kernel TOTAL"] = s1[:, 2] - s1[:, 0]
            d["feat tile load + geo search"] = st[:, 6] - st[:, 0]
        d["bin+slots"] = st[:, 7] - st[:, 6]
        d["s2 copies"] = st[:, 8]
        d["s2 lookups"] = st[:, 9]
        d["s2 total"] = st[:, 10] - st[:, 7]
        d["finalise+geo"] = st[:, 11] - st[:, 10]
        d["stage3"] = st[:, 12] - st[:, 11]
        blkt = st[:, 15] > 0                                # tiles that took the block tasks of stage 3
        d["s3 tasks (block tiles)"] = np.where(blkt, st[:, 15] - st[:, 11], np.nan)
        d["s3 ties+flush (block tiles)"] = np.where(blkt, st[:, 12] - st[:, 15], np.nan)
        d["stage3 (other tiles)"] = np.where(~blkt, st[:, 12] - st[:, 11], np.nan)
        d["sr_fused_kernel TOTAL" if not single else "TOTAL"] = st[:, 12] - st[:, 0]
        tot = st[:, 12] - st[:, 0]
        rt = st[:, 14] - st[:, 13]                       # s_memrealtime ticks (100 MHz) over the same interval as TOTAL
        mhz = tot / np.maximum(rt, 1) * 100.0
        wall = (st[:, 14].max() - st[:, 13].min()) / 100.0          # us, first tile start -> last tile end
        report(kind, d, cls, mhz, wall, "single launch" if single else "two launches (the headline path)")


if __name__ == "__main__":
    main()
