"""cProfile of one call-site frame (lazy + asdevice line): where the HOST time of the unchanged call sites goes."""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import lazy, luts as lutmod
    import callsite_driver as cd
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (1080, 1920, 3)).astype(np.float32)
    luts = cd.float_luts(lutmod.load_lut_arrays(os.path.join(lutmod.ASSET_DIR, "lerf-g")))
    interp, pads, resizer = cd.mirror_api(linear=False, support=2, max_sigma=10)

    def frame():
        x = lazy.asdevice(img)
        return np.asarray(cd.worker_sr(interp, pads, resizer, luts, x, (2.0, 2.0)))
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        frame()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
