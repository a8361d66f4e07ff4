"""CPU: lerf_pytorch_amd.lazy.DeviceArray answers the numpy operations of the reference's call sites
(resample/eval_lut_sr.py:541-665) exactly as numpy does -- here on CPU tensors (the class only needs a torch tensor; on the
GPU box tests/test_gpu_callsite.py runs the same expressions on device tensors and the whole worker protocol against the
reference's md5s)."""
import numpy as np
import pytest
import torch
from PIL import Image

from lerf_pytorch_amd import lazy


def _eq(d, n):
    return isinstance(d, lazy.DeviceArray) and d.dtype == n.dtype and d.shape == n.shape and np.array_equal(np.asarray(d), n)


@pytest.fixture()
def pair():
    rng = np.random.default_rng(0)
    a = rng.integers(-2032, 2033, (9, 37, 41)).astype(np.float64) / 16.0
    return lazy.DeviceArray(torch.from_numpy(a)), a


def test_accumulate_average_round_like_the_call_sites(pair):
    A, a = pair
    B, b = lazy.DeviceArray(torch.from_numpy(a[::-1].copy())), a[::-1].copy()
    pd, pn = 0, 0
    pd += A
    pn += a
    pd += B
    pn = pn + b
    assert _eq(pd, pn)
    for avg, bias in ((3, 0), (12, 127)):
        d = np.round(np.clip((pd / avg) + bias, 0, 255)).astype(np.float32).transpose((1, 2, 0))
        n = np.round(np.clip((pn / avg) + bias, 0, 255)).astype(np.float32).transpose((1, 2, 0))
        assert _eq(d, n) and _eq(d / float(255), n / float(255))
        for r in range(4):
            dr, nr = np.rot90(d, r), np.rot90(n, r)
            assert _eq(dr, nr) and dr.shape == nr.shape
            for p in (1, 3):
                assert _eq(np.pad(dr, ((0, p), (0, p), (0, 0)), mode="edge").transpose((2, 0, 1)),
                           np.pad(nr, ((0, p), (0, p), (0, 0)), mode="edge").transpose((2, 0, 1)))


def test_division_is_ieee_not_a_reciprocal_multiply():
    n = (np.arange(0, 4000, dtype=np.float64) * 24.0 - 24000.0) / 16.0       # N / 16 with N = 48 k + 24: N / 48 is a tie of the round
    assert _eq(np.round(lazy.DeviceArray(torch.from_numpy(n)) / 3), np.round(n / 3))
    q = np.arange(256, dtype=np.float32)
    assert _eq(lazy.DeviceArray(torch.from_numpy(q)) / float(255), q / float(255))
    assert _eq(np.round(lazy.DeviceArray(torch.from_numpy(np.array([0.5, 1.5, 2.5, -0.5, 254.5])))), np.round(np.array([0.5, 1.5, 2.5, -0.5, 254.5])))


def test_indexing_final_conversion_and_pil(pair):
    A, a = pair
    idx = list(range(1, 10, 3))
    assert _eq(A[idx, :, :], a[idx, :, :]) and _eq(A[2], a[2])
    u = np.clip(np.round(A).transpose((1, 2, 0)), 0, 255).astype(np.uint8)
    un = np.clip(np.round(a).transpose((1, 2, 0)), 0, 255).astype(np.uint8)
    assert _eq(u, un)
    assert np.array_equal(np.array(Image.fromarray(u[:, :, :3])), un[:, :, :3])


def test_everything_else_is_numpys_own_result(pair):
    A, a = pair
    assert float(A.max()) == a.max() and np.allclose(np.mean(A), a.mean()) and np.array_equal(np.abs(A), np.abs(a))
    assert np.array_equal(np.dot(A[0], np.ones(41)), np.dot(a[0], np.ones(41)))
    assert np.array_equal(A == 5.0, a == 5.0) and np.array_equal(np.asarray(A > a.mean()), a > a.mean())
    assert np.array_equal(A ** 2, a ** 2) and np.array_equal(np.isnan(A), np.isnan(a))
    np.testing.assert_allclose(A, a, rtol=0, atol=0)
    assert np.array_equal(np.float64(3.0) * A, 3.0 * a) and isinstance(3.0 / (A + 1000.0), lazy.DeviceArray)
    assert np.array_equal(np.concatenate([A, A], axis=0), np.concatenate([a, a], axis=0))
    assert np.array_equal(np.pad(A, 2, mode="reflect"), np.pad(a, 2, mode="reflect"))     # a mode the device path does not take


# ---- round 5: everything the reference's callers do to a result ----------------------
def test_torch_conversions_take_the_host_copy_without_a_python_walk():
    """resample/eval_lut_warp.py:233 `torch.Tensor(img_out)`; torch.tensor / torch.as_tensor alike.  A DeviceArray is a CPU
    array to torch (DLPack export of the host copy); the conversion must not walk the elements in Python."""
    import time
    a = np.random.default_rng(1).integers(0, 256, (300, 400, 3), dtype=np.uint8)
    A = lazy.DeviceArray(torch.from_numpy(a.copy()))
    t0 = time.time()
    x, y, z = torch.Tensor(A), torch.tensor(A), torch.as_tensor(A)
    assert time.time() - t0 < 1.0
    assert x.dtype == torch.float32 and torch.equal(x, torch.Tensor(a)) and x.device.type == "cpu"
    assert y.dtype == torch.uint8 and torch.equal(y, torch.tensor(a)) and torch.equal(z, torch.as_tensor(a))
    y[0, 0, 0] += 1                                            # a private copy: the array itself is untouched
    assert np.array_equal(np.asarray(A), a)
    B = lazy.DeviceArray(torch.from_numpy(np.array(a == 7)))
    assert torch.equal(torch.Tensor(np.array(B)), torch.Tensor(np.array(a == 7)))
    s = lazy.DeviceArray(torch.tensor(3.5, dtype=torch.float64))
    with pytest.raises(TypeError):
        len(s)
    assert float(torch.tensor(s)) == 3.5 and float(s) == 3.5


def test_item_assignment_and_views_share_storage_like_numpy(pair):
    A, a = pair
    a = a.copy()
    A[1] = 5.0
    a[1] = 5.0
    A[:, 2:4, ::3] = np.float32(-1)
    a[:, 2:4, ::3] = np.float32(-1)
    m = a > 100.0
    A[m] = 0
    a[m] = 0
    A[lazy.DeviceArray(torch.from_numpy(a < -100.0))] = 7
    a[a < -100.0] = 7
    A[[0, 3], :, :] = a[[1, 2], :, :] * 2
    a[[0, 3], :, :] = a[[1, 2], :, :] * 2
    assert _eq(A, a)
    v, vn = A.transpose((1, 2, 0)), a.transpose((1, 2, 0))
    hv = np.asarray(v)                                        # cached host copy of the view ...
    v[3:5, 1] = 99.0                                          # ... a write through the view reaches the base and drops the caches
    vn[3:5, 1] = 99.0
    assert _eq(A, a) and _eq(v, vn) and not hv.flags.writeable
    col, coln = A.reshape(9, -1)[:, 0], a.reshape(9, -1)[:, 0]
    col += 1000.0                                             # `t[:, 0] += O[0]` of the colour transform (common/utils.py:68-70)
    coln += 1000.0
    assert _eq(A, a)
    u = lazy.DeviceArray(torch.zeros((4, 5), dtype=torch.uint8))
    un = np.zeros((4, 5), dtype=np.uint8)
    u[1:3, 1:4] = 255
    un[1:3, 1:4] = 255
    u[0] = np.arange(5)
    un[0] = np.arange(5)
    assert _eq(u, un)
    for bad in (300, -1):
        with pytest.raises(OverflowError):
            u[0, 0] = bad
    assert A[1, 2, 3] == a[1, 2, 3] and isinstance(A[1, 2, 3], np.float64)


def test_inplace_operators_mutate_like_numpy(pair):
    A, a = pair
    a = a.copy()
    B, b = A, a
    A += 2
    a += 2
    A *= 0.5
    a *= 0.5
    A -= a[0]
    a -= a[0].copy()
    A /= 3
    a /= 3
    assert B is A and _eq(B, a) and b is a
    A -= A[0]                                                 # an operand that overlaps the destination: read before written, like numpy
    a -= a[0]
    A *= A.transpose((0, 1, 2))[:, :1, :]
    a *= a[:, :1, :]
    assert _eq(A, a)
    i = lazy.DeviceArray(torch.arange(6, dtype=torch.int32))
    n = np.arange(6, dtype=np.int32)
    with pytest.raises(TypeError):
        n /= 2
    with pytest.raises(TypeError):
        i /= 2
    with pytest.raises(TypeError):
        i += 0.5
    p = np.zeros(a.shape)
    p += A                                                    # an ndarray accumulator takes the host path: same values
    assert np.array_equal(p, a) and type(p) is np.ndarray


def test_result_dtypes_are_numpys(pair):
    A, a = pair
    F, f = lazy.DeviceArray(torch.from_numpy(a.astype(np.float32))), a.astype(np.float32)
    I, i = lazy.DeviceArray(torch.from_numpy((a * 16).astype(np.int32))), (a * 16).astype(np.int32)
    U, u = lazy.DeviceArray(torch.from_numpy((a % 256).astype(np.uint8))), (a % 256).astype(np.uint8)
    cases = [(I * 0.5, i * 0.5), (I / 4, i / 4), (F + np.float64(3.0), f + np.float64(3.0)), (F * 0.1, f * 0.1), (F + 1, f + 1),
             (F + a, f + a), (2 - F, 2 - f), (1.0 / (F + 3000), 1.0 / (f + 3000)), (I + I, i + i), (U * (u > 3), u * (u > 3)),
             (U + I, u + i), (F / np.float32(255), f / np.float32(255)), (-I, -i), (abs(I), abs(i)), (U + 3, u + 3),
             (F == 5, f == 5), (I == 7, i == 7), (U != 255, u != 255), (np.less(0.0, F), np.less(0.0, f)),
             (F.clip(0, 255), f.clip(0, 255)), (np.clip(I, 0, 255), np.clip(i, 0, 255)), (np.clip(U, 0.5, 3), np.clip(u, 0.5, 3)),
             (F.astype(np.uint8), f.astype(np.uint8)), (A.astype(np.int16), a.astype(np.int16)), (np.round(I), np.round(i))]
    for k, (d, n) in enumerate(cases):
        assert d.dtype == n.dtype and d.shape == n.shape and np.array_equal(np.asarray(d), n), k
    with pytest.raises(OverflowError):
        u + 300
    with pytest.raises(OverflowError):
        U + 300
    assert np.array_equal(np.asarray(U.clip(None, None)), u.clip(None, None)) if hasattr(np, "clip") else True


def test_out_arguments_write_through_to_the_device(pair):
    A, a = pair
    a = a.copy()
    O, o = lazy.DeviceArray(torch.zeros(a.shape, dtype=torch.float64)), np.zeros(a.shape)
    r = np.add(A, 1.0, out=O)
    np.add(a, 1.0, out=o)
    assert r is O and _eq(O, o) and torch.equal(O.t, torch.from_numpy(o))       # the device tensor itself, not a stale cache
    r = np.clip(A, -5, 5, out=O)
    np.clip(a, -5, 5, out=o)
    assert r is O and torch.equal(O.t, torch.from_numpy(o))
    np.multiply(a, 2.0, out=O)                                # ndarray inputs, DeviceArray out
    assert torch.equal(O.t, torch.from_numpy(a * 2.0))
    h = np.asarray(A)
    with pytest.raises(ValueError):
        h[0, 0, 0] = 1.0                                      # the cached host copy is read-only ...
    w = np.array(A)
    w[0, 0, 0] = 1.0                                          # ... np.array gives a private writable one
    assert np.array_equal(np.asarray(A), a)


def test_upload_size_classes():
    assert lazy._size_class(1 << 20) == 1 << 20 and lazy._size_class((1 << 20) + 1) == (1 << 20) + (1 << 18)
    for n in (1080 * 1920 * 3 * 4, 1081 * 1925 * 3 * 4, 25_000_001, 12_345_678):
        c = lazy._size_class(n)
        assert n <= c <= n * 1.25 + 4096
    assert len({lazy._size_class(n) for n in range(24_000_000, 26_000_000, 7919)}) <= 2


@pytest.mark.parametrize("wrap", [True, False])
def test_worker_tails_on_device_arrays_equal_the_tails_on_numpy(tmp_path, oracle, luts_g, wrap):
    """tools/callsite_driver.py tail_sr / tail_warp (resample/eval_lut_sr.py:667-744, eval_lut_warp.py:221-302) on
    DeviceArray results (CPU tensors here; the GPU suite runs the whole protocol against the reference's numbers): the files
    they write and the metrics they return are those of the same statements on numpy arrays."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import callsite_driver as cd
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)
    gt = rng.integers(0, 256, (81, 95, 3), dtype=np.uint8)               # the crop of :735-739 in both directions
    feat, hq, o64, o8 = oracle.sr_pipeline(img, luts_g, 2, 2, return_all=True)
    feat_chw = feat.astype(np.float32).transpose((2, 0, 1))
    hyper = (hq.astype(np.float32) / 255.0).transpose((2, 0, 1)) if hq.ndim == 3 and hq.shape[2] == 9 else hq.astype(np.float32) / 255.0
    mask_chw = np.where(rng.random((3, 81, 95)) < 0.8, 255.0, rng.choice([0.0, 127.5], (3, 81, 95)))
    out_w = rng.integers(0, 256, (81, 95, 3), dtype=np.uint8)
    W = (lambda x: lazy.DeviceArray(torch.from_numpy(np.ascontiguousarray(x)))) if wrap else (lambda x: x)
    d = tmp_path / ("dev" if wrap else "np")
    d.mkdir()
    r_sr = cd.tail_sr(W(o8), W(feat_chw), W(hyper), gt, (2, 2), str(d), "img")
    r_wp = cd.tail_warp(W(out_w), W(mask_chw), W(feat_chw), gt, str(d), "wimg")
    ref = tmp_path / "ref"
    ref.mkdir()
    e_sr = cd.tail_sr(o8, feat_chw, hyper, gt, (2, 2), str(ref), "img")
    e_wp = cd.tail_warp(out_w, mask_chw, feat_chw, gt, str(ref), "wimg")
    assert [float(v) for v in r_sr] == [float(v) for v in e_sr] and float(r_wp[0]) == float(e_wp[0])
    names = sorted(os.listdir(ref))
    assert names == sorted(os.listdir(d)) and len(names) == 8
    for n in names:
        assert open(os.path.join(d, n), "rb").read() == open(os.path.join(ref, n), "rb").read(), n


def test_corner_cases_of_the_ndarray_surface():
    """empty and 0-dim arrays, negative-step slices (torch refuses them: host copy), boolean / integer index arrays, iteration,
    numpy functions that take sequences of arrays, deep copies"""
    import copy
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    A = lazy.DeviceArray(torch.from_numpy(a.copy()))
    E = lazy.DeviceArray(torch.empty((0, 3), dtype=torch.float64))
    assert len(E) == 0 and np.asarray(E).shape == (0, 3) and (E + 1).shape == (0, 3) and torch.tensor(E).shape == (0, 3)
    Z = lazy.DeviceArray(torch.tensor(2.5, dtype=torch.float64))
    assert float(Z) == 2.5 and float(Z / 3) == 2.5 / 3 and bool(Z > 1) and np.asarray(Z * 2) == 5.0
    assert np.array_equal(np.asarray(A[::-1]), a[::-1]) and np.array_equal(np.asarray(A[:, ::-2]), a[:, ::-2])
    B, b = A.copy(), a.copy()
    B[::-1, 0] = [3, 4]
    b[::-1, 0] = [3, 4]
    B[...] = B + 1
    b[...] = b + 1
    assert _eq(B, b)
    assert _eq(A[[True, False]], a[[True, False]]) and _eq(A[np.array([1, 0]), 1:], a[np.array([1, 0]), 1:]) and _eq(A[A > 2], a[a > 2])
    assert A[..., None].shape == (2, 3, 1) and A[None].shape == (1, 2, 3) and _eq(A[-1], a[-1]) and _eq(A.T, a.T)
    assert [list(map(float, r)) for r in A] == a.tolist()
    assert np.array_equal(np.stack([A, A]), np.stack([a, a])) and np.array_equal(np.where(A > 2, A, 0), np.where(a > 2, a, 0))
    C = copy.deepcopy(A)
    C += 1
    assert _eq(A, a) and _eq(C, a + 1)
