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
