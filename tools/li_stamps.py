"""Timeline of the LDS-resident LUT pass from a stamped diagnostic build (tools/build_li_variant.sh stamps "-DLERF_LI_STAMPS"):
s_memrealtime stamps (10 ns) per workgroup: 0 entry, 1 LUT staged, 2 first tile staged, then compute / commit pairs.
Usage: python tools/li_stamps.py --lib PATH [--oc 1] [--mode s] [--rot 0]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--oc", type=int, default=1)
    ap.add_argument("--mode", default="s")
    ap.add_argument("--rot", type=int, default=0)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--kernel", default="lds")
    a = ap.parse_args()
    import torch
    from lerf_pytorch_amd import _lib, ops
    _lib.use_library(a.lib)
    L = _lib.lib()
    rng = np.random.default_rng(0)
    lut = torch.from_numpy(rng.integers(-128, 128, (17 ** 4, a.oc), dtype=np.int8)).cuda()
    h, w = (a.h, a.w) if a.rot % 2 == 0 else (a.w, a.h)
    x = torch.from_numpy(rng.integers(0, 256, (h + 3, w + 3, 3), dtype=np.uint8)).cuda().to(torch.float32).permute(2, 0, 1)
    dy, dx = _lib.mode_offsets(a.mode, 0)
    out = torch.empty((3 * a.oc,) + ((h, w) if a.rot % 2 == 0 else (w, h)), dtype=torch.float64, device="cuda")
    st = torch.zeros((256, 32), dtype=torch.int64, device="cuda")
    for _ in range(5):
        ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=a.rot, out=out, kernel=a.kernel)
    torch.cuda.synchronize()
    L.lerf_li_set_stamps.argtypes = [C.c_void_p]
    L.lerf_li_set_stamps.restype = None
    L.lerf_li_set_stamps(st.data_ptr())
    ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=a.rot, out=out, kernel=a.kernel)
    torch.cuda.synchronize()
    L.lerf_li_set_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    t0 = s[:, 0][s[:, 0] > 0].min()
    n = int((s > 0).sum(axis=1).max())
    print("oC %d mode %s rot %d: stamps (us after the first workgroup's entry), median / min / max over workgroups" % (a.oc, a.mode, a.rot))
    names = ["entry", "lut staged", "tile 0 staged"]
    for k in range(32):
        col = s[:, k]
        col = (col[col > 0] - t0) / 100.0
        if col.size == 0:
            continue
        if k >= 14:
            nm = "loader tile %d: %s" % ((k - 14) // 3 + 1, ("loads landed", "committed", "next issued")[(k - 14) % 3])
        else:
            nm = names[k] if k < 3 else ("compute %d done" % ((k - 3) // 2) if (k - 3) % 2 == 0 else "tile %d staged" % ((k - 3) // 2 + 1))
        print("  %2d %-32s n=%3d  median %7.2f  min %7.2f  max %7.2f" % (k, nm, col.size, np.median(col), col.min(), col.max()))


if __name__ == "__main__":
    main()
