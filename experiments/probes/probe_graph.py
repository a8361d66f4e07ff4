#!/usr/bin/env python3
"""Is the fused path capturable in a HIP graph (torch.cuda.CUDAGraph), and what does replay save on launch-bound sizes?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
rng = np.random.default_rng(0)
for (model, S, N, H, W, sc) in [("lerf-g", 2, 1, 256, 256, 2), ("lerf-g", 2, 1, 96, 128, 3), ("lerf-g", 4, 1, 256, 256, 2), ("lerf-l", 2, 1, 256, 256, (1.5, 2.0)),
                                ("lerf-g", 2, 1, 1080, 1920, 2), ("lerf-g", 2, 8, 1080, 1920, 2)]:
    eng = L.LerfEngine.shipped(model, support=S)
    x = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    for _ in range(3):
        want = eng.sr(x, sc)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            eng.sr(x, sc)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            y = eng.sr(x, sc)
    except Exception as e:
        print("capture FAILED", model, S, (N, H, W), repr(e)[:300]); continue
    x2 = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    x.copy_(x2)
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(y, eng.sr(x2, sc))
    def T(f, n=200):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    te = T(lambda: eng.sr(x, sc)); tg = T(g.replay)
    print("%s S=%d %dx%dx%d x%s: eager %.4f ms, graph replay %.4f ms per call, bytes after replay on new input equal: %s" % (model, S, N, H, W, sc, te, tg, ok))
