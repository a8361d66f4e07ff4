"""CPU (gloo, world 2 / 4): the partition of a homographic warp by output rows (dist.WarpRowPlan).  Every rank gets only its band
of the input -- the rows outside it are POISONED -- runs the LUT stages on the band and the warp with the whole frame's
geometry on its output rows; the gathered rows are the whole-frame result.  The per-rank compute is the ORACLE (the checker);
the product kernels run the same plan in tests/test_gpu_warp_fused.py::test_warp_row_partition_single_gpu."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lerf_pytorch_amd  # noqa: F401
from lerf_pytorch_amd import dist as ldist

M_ISC = [[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]]
M_ROT = [[0.9, -1.7, 90.0], [1.6, 1.1, -20.0], [2e-4, 4e-4, 1.0]]
M_OSC = [[4.1, 0.4, 30.0], [0.5, 3.8, 25.0], [8e-5, 1.2e-4, 1.0]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, M, out_hw, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import lerf_oracle as O
    from conftest import ASSETS
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                      # the same frame on every rank
    plan = ldist.WarpRowPlan(H, W, M, out_hw, world, rank, 2)
    luts = O.load_luts(os.path.join(ASSETS, "lerf-g"))
    band = img[plan.b0:plan.b1]
    feat_b, hq_b = O.lut_stages(band, luts, 3)
    poison = np.random.default_rng(100 + rank)
    feat = poison.integers(0, 256, (H, W, 3), dtype=np.uint8)                  # what the rank does not hold is garbage
    hq = poison.integers(0, 256, (H, W, 3, 3), dtype=np.uint8)
    # the stage outputs of the band are the frame's except within 6 rows of an artificial band edge
    lo = plan.b0 if plan.b0 == 0 else plan.b0 + 6
    hi = plan.b1 if plan.b1 == H else plan.b1 - 6
    feat[lo:hi], hq[lo:hi] = feat_b[lo - plan.b0:hi - plan.b0], hq_b[lo - plan.b0:hi - plan.b0]
    assert lo <= plan.t0 and plan.t1 <= hi
    mine = O.to_u8(np.nan_to_num(O.warp_u8(feat, hq, np.array(M), out_hw, 2, 10, "gauss"), nan=0.0))[plan.i0:plan.i1]
    counts = [ldist.WarpRowPlan(H, W, M, out_hw, world, r, 2).out_rows() for r in range(world)]
    whole = ldist.gather_strips(torch.from_numpy(mine.copy()), counts)
    np.save(os.path.join(tmp, "whole_%d.npy" % rank), whole.numpy())
    np.save(os.path.join(tmp, "band_%d.npy" % rank), np.array([plan.b0, plan.b1, plan.i0, plan.i1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,M,out_hw,world", [(60, 50, M_ISC, (110, 96), 2), (64, 48, M_ROT, (90, 100), 4), (50, 60, M_OSC, (100, 120), 2)])
def test_gloo_warp_rows_stitch_to_the_whole_frame(tmp_path, oracle, luts_g, H, W, M, out_hw, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, M, out_hw, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    full = oracle.warp_pipeline(img, luts_g, np.array(M), out_hw)
    rows = 0
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("whole_%d.npy" % r)), full)
        b0, b1, i0, i1 = np.load(tmp_path / ("band_%d.npy" % r))
        assert 0 <= b0 < b1 <= H and i0 == rows
        rows = i1
    assert rows == out_hw[0]


def test_warp_row_plan_bands_are_smaller_than_the_frame_for_in_scale_warps():
    """the 8-rank partition of BASELINE config 4 (1080p -> 4K, isc-like matrix): every rank holds an eighth of the rows + the skew of
    the matrix over the frame's width (0.08 x 1920 / 1.95 rows) + the reach of the LUT stages -- a quarter of the frame at most"""
    bands = [ldist.WarpRowPlan(1080, 1920, M_ISC, (2160, 3840), 8, r, 2) for r in range(8)]
    assert bands[0].b0 == 0 and bands[-1].b1 == 1080
    assert all(b.b1 - b.b0 <= 1080 // 4 for b in bands)
    assert all(bands[k].i1 == bands[k + 1].i0 for k in range(7)) and bands[0].i0 == 0 and bands[-1].i1 == 2160
