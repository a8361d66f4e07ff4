"""Run under AddressSanitizer + UBSan (tests/test_sanitizers_cpu.py starts it with libasan preloaded): the host-only entry points
of include/lerf_hip.h from the sanitizer build (csrc/build_asan/liblerf_host_asan.so, plain g++) against the product library
(liblerf_hip.so, hipcc) on the geometry the suite and the benchmarks use plus randomised sizes, scales, supports and
homographies -- bit-equal tables, and no sanitizer report."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def bind(L):
    L.lerf_sr_axis_tables.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_sr_axis_tables_f32.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lerf_out_size.argtypes = [C.c_int, C.c_double]
    L.lerf_invert3x3.argtypes = [C.c_void_p, C.c_void_p]
    L.lerf_warp_pads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lerf_mode_offsets.argtypes = [C.c_char, C.c_int, C.c_void_p, C.c_void_p]
    return L


def tables(L, n_in, n_out, s, S, f32):
    left = np.zeros(n_out, np.int32)
    pads = np.zeros(2, np.int32)
    if f32:
        d = np.zeros(n_out * S, np.float32)
        rc = L.lerf_sr_axis_tables_f32(n_in, n_out, s, S, left.ctypes.data, d.ctypes.data, pads.ctypes.data)
        return rc, left, d, pads
    d64, d32 = np.zeros(n_out * S, np.float64), np.zeros(n_out * S, np.float32)
    rc = L.lerf_sr_axis_tables(n_in, n_out, s, S, left.ctypes.data, d64.ctypes.data, d32.ctypes.data, pads.ctypes.data)
    return rc, left, d64, d32, pads


def main():
    san = bind(C.CDLL(sys.argv[1]))
    prod = bind(C.CDLL(os.path.join(REPO, "lerf-pytorch_amd", "liblerf_hip.so")))
    assert san.lerf_abi_version() == prod.lerf_abi_version()
    rng = np.random.default_rng(0)
    cases = [(1080, 2.0, 2), (1920, 2.0, 4), (1080, 1.5, 2), (2160, 2.0, 2), (5, 2.4, 2), (6, 1.3, 2), (17, 3.0, 2), (16, 4.0, 8), (1, 1.0, 1), (7, 0.5, 4)]
    cases += [(int(rng.integers(1, 3000)), float(rng.uniform(0.3, 8.0)), int(rng.integers(1, 9))) for _ in range(300)]
    n = 0
    for n_in, s, S in cases:
        n_out = san.lerf_out_size(n_in, s)
        assert n_out == prod.lerf_out_size(n_in, s)
        if n_out < 1:
            continue
        for f32 in (False, True):
            a, b = tables(san, n_in, n_out, s, S, f32), tables(prod, n_in, n_out, s, S, f32)
            assert a[0] == b[0] == 0, (n_in, s, S, a[0], b[0])
            for x, y in zip(a[1:], b[1:]):
                assert x.tobytes() == y.tobytes(), (n_in, s, S, f32)
            n += 1
    # argument checks (no write through a bad argument)
    assert san.lerf_sr_axis_tables(0, 4, 2.0, 2, None, None, None, None) == -1
    assert san.lerf_sr_axis_tables(4, 8, 2.0, 9, None, None, None, None) == -1
    for _ in range(200):
        M = np.eye(3) + rng.normal(0, 0.3, (3, 3))
        M[2, :2] *= 1e-3
        M[:2, 2] *= 50
        inv_s, inv_p = np.zeros(9), np.zeros(9)
        ra, rb = san.lerf_invert3x3(np.ascontiguousarray(M).ctypes.data, inv_s.ctypes.data), prod.lerf_invert3x3(np.ascontiguousarray(M).ctypes.data, inv_p.ctypes.data)
        assert ra == rb and inv_s.tobytes() == inv_p.tobytes()
        if ra != 0:
            continue
        H, W, oH, oW, S = (int(v) for v in (rng.integers(4, 2200), rng.integers(4, 4000), rng.integers(4, 4400), rng.integers(4, 8000), rng.integers(1, 9)))
        ps, pp = np.zeros(4, np.int32), np.zeros(4, np.int32)
        assert san.lerf_warp_pads(inv_s.ctypes.data, H, W, oH, oW, S, ps.ctypes.data) == prod.lerf_warp_pads(inv_p.ctypes.data, H, W, oH, oW, S, pp.ctypes.data) == 0
        assert ps.tobytes() == pp.tobytes()
    for mode in b"sdyctq":
        for rot in range(-2, 7):
            d1, d2 = np.zeros(8, np.int8), np.zeros(8, np.int8)
            m = C.c_char(bytes([mode]))
            ra, rb = san.lerf_mode_offsets(m, rot, d1.ctypes.data, d1.ctypes.data + 4), prod.lerf_mode_offsets(m, rot, d2.ctypes.data, d2.ctypes.data + 4)
            assert ra == rb and (ra != 0 or d1.tobytes() == d2.tobytes())
    print("sanitized host functions ok: %d table pairs, 200 homographies" % n)


if __name__ == "__main__":
    main()
