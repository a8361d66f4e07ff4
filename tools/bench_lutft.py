#!/usr/bin/env python3
"""Throughput of one LUT fine-tuning iteration (train_model.py:416-442 with the reference's defaults: batch 256 from
scripts.sh:30, 48x48 LR patches, x4, Adam) on the MI355X path: SWF2LUT.predict (24 HIP LUT passes + torch glue),
SteeringGaussianResize2dTorch forward/backward (HIP), MSE, optimiser step.   usage: bench_lutft.py [batch] [scale]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd  # noqa: F401
from lerf_pytorch_amd.resample.model import SWF2LUT, lutft_step
from lerf_pytorch_amd.resize_right.resize_right2d_torch import AmplifiedLinearResize2dTorch, SteeringGaussianResize2dTorch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
for name, linear in (("lerf-g", False), ("lerf-l", True)):
    opt = types.SimpleNamespace(modes="sct", modes2="sct", stages=2, norm=255, interval=4, lutName="LUTft",
                                expDir=os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", name))
    m = SWF2LUT(opt, inC=1, outC=1 if linear else 3).cuda()
    r = (AmplifiedLinearResize2dTorch(support_sz=2, device=torch.device("cuda")) if linear else
         SteeringGaussianResize2dTorch(support_sz=2, device=torch.device("cuda"), max_sigma=10))
    r.set_shape([B, 1, 48, 48], scale_factors=scale)
    rng = np.random.default_rng(0)
    if os.environ.get("LUTFT_SMOOTH"):          # photo-like patches: strong LUT-row locality, the hard case for the atomics
        base = rng.random((B, 1, 6, 6), dtype=np.float32)
        hr = np.kron(base, np.ones((1, 1, int(8 * scale), int(8 * scale)), np.float32))
        hr = (0.25 * (hr + np.roll(hr, 5, 2) + np.roll(hr, 9, 3) + np.roll(np.roll(hr, 3, 2), 7, 3))).astype(np.float32)
        S = int(scale)
        lr = hr.reshape(B, 1, 48, S, 48, S).mean(axis=(3, 5))
        im, lb = torch.tensor(lr, device="cuda"), torch.tensor(hr, device="cuda")
    else:
        im = torch.tensor(rng.random((B, 1, 48, 48), dtype=np.float32), device="cuda")
        lb = torch.tensor(rng.random((B, 1, int(48 * scale), int(48 * scale)), dtype=np.float32), device="cuda")
    opt_G = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    for _ in range(3): lutft_step(m, r, im, lb, opt_G, linear=linear)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): lutft_step(m, r, im, lb, opt_G, linear=linear)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("%s fine-tuning step, batch %d x 48x48 -> x%g: %.2f ms/iteration = %.0f patches/s" % (name, B, scale, dt * 1e3, B / dt))
