"""Times one FourSimplexInterpFaster-shaped call (lerf_lut_interp_ex) at 1080p: float32 HWC operand viewed as CHW, float64 planes
out, per (oC, pattern, rotation), LDS-resident kernel beside the direct kernel, accumulate form beside the plain one.
Usage: python tools/bench_lut_interp.py [--h 1080 --w 1920] [--iters 20] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--lib", default=None, help="a variant build of liblerf_hip.so (tools/build_li_variant.sh)")
    ap.add_argument("--kernels", default="lds,direct")
    ap.add_argument("--oc", default="1,3")
    ap.add_argument("--no-acc", action="store_true")
    ap.add_argument("--planes", action="store_true", help="hand the LDS kernel the plane form of the LUT")
    a = ap.parse_args()
    import torch
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import _lib, ops
    if a.lib:
        _lib.use_library(a.lib)
    rng = np.random.default_rng(0)
    rows = []
    for oC in [int(v) for v in a.oc.split(",")]:
        lut = torch.from_numpy(rng.integers(-128, 128, (17 ** 4, oC), dtype=np.int8)).cuda()
        planes = ops.lut_planes(lut) if a.planes and oC > 1 else None
        for mode in ("s", "c", "t"):
            for rot in (0, 1, 2, 3):
                if a.quick and not ((mode, rot) in (("s", 0), ("c", 1), ("t", 2))):
                    continue
                h, w = (a.h, a.w) if rot % 2 == 0 else (a.w, a.h)
                hwc = torch.from_numpy(rng.integers(0, 256, (h + 3, w + 3, 3), dtype=np.uint8)).cuda().to(torch.float32)
                x = hwc.permute(2, 0, 1)
                dy, dx = _lib.mode_offsets(mode, 0)
                out = torch.empty((3 * oC,) + ((h, w) if rot % 2 == 0 else (w, h)), dtype=torch.float64, device="cuda")
                row = {"oC": oC, "mode": mode, "rot": rot}
                if rot == 0 and mode == "s":
                    # the practical floor of the call's stream on this chip: writing the output planes alone (a device fill)
                    for _ in range(3):
                        out.fill_(0.5)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.iters):
                        out.fill_(0.5)
                    e1.record()
                    torch.cuda.synchronize()
                    row["fill_output_us"] = round(e0.elapsed_time(e1) * 1000.0 / a.iters, 2)
                    row["fill_output_tb_s"] = round(out.numel() * 8 / (row["fill_output_us"] * 1e6), 2)
                for kern in a.kernels.split(","):
                    for acc in ((False,) if a.no_acc else (False, True)):
                        if acc:
                            out.zero_()
                        for _ in range(3):
                            ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, out=out, accumulate=acc, kernel=kern, planes=None if kern == "direct" else planes)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize()
                        e0.record()
                        for _ in range(a.iters):
                            ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, out=out, accumulate=acc, kernel=kern, planes=None if kern == "direct" else planes)
                        e1.record()
                        torch.cuda.synchronize()
                        row["%s%s_us" % (kern, "_acc" if acc else "")] = round(e0.elapsed_time(e1) * 1000.0 / a.iters, 2)
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
