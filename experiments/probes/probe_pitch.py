import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import dist as ldist, ops
H, W, scale, grid, S = 200, 260, 2, (2, 4), 2
eng = L.LerfEngine.shipped("lerf-g", support=S)
img = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
x = torch.from_numpy(img).cuda()
full = eng.sr(x, scale)
geo = eng.sr_geometry((H, W), scale)
lr, lc = geo.host["left_r"], geo.host["left_c"]
for r in range(8):
    p = ldist.BlockPlan(H, W, grid, r, S, lr, lc)
    ext = x[p.ylo:p.yhi, p.xlo:p.xhi].contiguous().unsqueeze(0)
    i0, i1, j0, j1 = p.out_rect()
    want = full[i0:i1, j0:j1]
    lg = ldist.block_geometry(geo, p)
    for name, kw in (("dense 1-launch", dict(workspace=False)), ("pitched 1-launch", dict(workspace=False, out=ldist.block_output(p, 1, 3, ext.device))),
                     ("dense 2-launch", dict()), ("pitched 2-launch", dict(out=ldist.block_output(p, 1, 3, ext.device)))):
        got = ops.sr_fused_u8(ext, eng.luts, lg, eng.kind, eng.max_sigma, **kw)[0]
        d = (got != want).any(dim=2)
        if d.any():
            ys, xs = torch.nonzero(d, as_tuple=True)
            print("rank %d %s: %d px differ, rows %d..%d cols %d..%d of %s; roi %s local %s" % (r, name, int(d.sum()), ys.min(), ys.max(), xs.min(), xs.max(), tuple(want.shape), p.roi, p.local_hw))
        else:
            print("rank %d %s: ok" % (r, name))
