"""Deterministic inputs of the g10 (SWF2LUT) golden cases, shared by gen_golden.py (which feeds them to the
reference) and the tests (which feed them to the HIP path).  numpy Generator streams are stable across runs."""
import numpy as np

MODE_PAD = {"s": 1, "d": 2, "y": 2, "c": 3, "t": 3}


def case_inputs(seed, mode, outC, B=2, Cn=2, h=7, w=9):
    rng = np.random.default_rng(seed)
    bd = MODE_PAD[mode]
    img = rng.integers(0, 256, (B, Cn, h + bd, w + bd)).astype(np.float32)
    G = rng.standard_normal((B, Cn * outC, h, w)).astype(np.float32)
    return bd, img, G


def case_weight(base, seed):
    """base: float32 [17^4, oC] = LUT / 127.  Moves it off the 1/127 grid and pushes some rows beyond the clamp."""
    rng = np.random.default_rng(seed)
    w = base.astype(np.float32).copy()
    w += rng.standard_normal(w.shape).astype(np.float32) * np.float32(0.002)
    w[rng.integers(0, w.shape[0], 4000)] *= np.float32(1.5)
    return w
