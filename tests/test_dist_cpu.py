"""CPU tests of the multi-GPU strip partition (gloo, world_size 2): halo exchange
delivers exactly the rows each rank needs, and stitching per-strip results
reproduces the full frame.  The per-strip compute here is the ORACLE (the
checker) -- the product compute path needs a GPU and is covered by
tests/test_gpu_parity.py::test_strip_partition_single_gpu."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lerf_pytorch_amd  # noqa: F401
from lerf_pytorch_amd import dist as ldist


@pytest.mark.parametrize("H,world,S,scale", [(64, 2, 2, 2.0), (90, 4, 2, 1.5), (80, 3, 4, 3.0), (1080, 8, 2, 2.0),
                                             (2160, 8, 2, 2.0), (77, 2, 2, 2.4)])
def test_plan_covers_rows_and_support(oracle, H, world, S, scale):
    left, _, _, _ = oracle.sr_axis_tables(H, oracle.out_size(H, scale), scale, S)
    outs = []
    lr_rows = []
    for r in range(world):
        p = ldist.StripPlan(H, world, r, S, left)
        assert p.halo == 3 + 3 + S // 2
        assert p.check_support(left)
        outs.append(p.out_rows())
        lr_rows.append(p.owned())
    assert lr_rows[0][0] == 0 and lr_rows[-1][1] == H
    assert all(lr_rows[i][1] == lr_rows[i + 1][0] for i in range(world - 1))
    assert outs[0][0] == 0 and outs[-1][1] == len(left)
    assert all(outs[i][1] == outs[i + 1][0] for i in range(world - 1))


def test_strips_thinner_than_the_halo_are_refused(oracle):
    left, _, _, _ = oracle.sr_axis_tables(20, 40, 2.0, 2)
    with pytest.raises(ValueError, match="thinner than the halo"):
        ldist.StripPlan(20, 4, 1, 2, left)
    ldist.StripPlan(20, 1, 0, 2, left)          # a world of one has no halo to exchange


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, scale, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import lerf_oracle as O
    from conftest import ASSETS
    rng = np.random.default_rng(42)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)        # same frame on every rank
    left, _, _, _ = O.sr_axis_tables(H, O.out_size(H, scale), scale, 2)
    plan = ldist.StripPlan(H, world, rank, 2, left)
    own = torch.from_numpy(img[plan.y0:plan.y1].copy())
    ext = ldist.exchange_halos(own, plan)
    ok_halo = np.array_equal(ext.numpy(), img[plan.ylo:plan.yhi])
    # batched exchange through a persistent StripBuffer: rows are produced straight into `own`, halos arrive in place
    buf = ldist.StripBuffer(plan, 2, W, 3)
    for rep in range(2):                                          # reused across steps
        buf.own.copy_(torch.stack([own, own]))
        ext2 = buf.exchange()
        ok_halo = ok_halo and ext2.data_ptr() == buf.ext.data_ptr()
        ok_halo = ok_halo and np.array_equal(ext2[1].numpy(), img[plan.ylo:plan.yhi])
    # per-strip compute with the checker: LUT stages on the strip alone (their values are wrong only in the outer
    # 6 rows of an artificial strip border, which stage 3 of the owned rows never reads), stage 3 with the GLOBAL
    # geometry -- the strip is pasted into an otherwise empty frame so that non-integer scales partition exactly
    luts = O.load_luts(os.path.join(ASSETS, "lerf-g"))
    feat_s, hq_s = O.lut_stages(ext.numpy(), luts, 3)
    feat = np.zeros((H, W, 3), np.uint8)
    hq = np.zeros((H, W, 3, 3), np.uint8)
    feat[plan.ylo:plan.yhi], hq[plan.ylo:plan.yhi] = feat_s, hq_s
    mine = O.to_u8(O.resize_u8(feat, hq, scale, scale))[plan.i0:plan.i1]     # global output rows [i0, i1)
    counts = [ldist.StripPlan(H, world, r, 2, left).out_rows() for r in range(world)]
    whole = ldist.gather_strips(torch.from_numpy(mine.copy()), counts)        # unequal strips, one equal-size all-gather
    np.save(os.path.join(tmp, "out_%d.npy" % rank), mine)
    np.save(os.path.join(tmp, "whole_%d.npy" % rank), whole.numpy())
    np.save(os.path.join(tmp, "ok_%d.npy" % rank), np.array([ok_halo, plan.check_support(left)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,scale,world", [(40, 24, 2.0, 2), (45, 20, 1.5, 2), (50, 16, 3.0, 3)])
def test_gloo_halo_exchange_and_stitch(tmp_path, oracle, luts_g, H, W, scale, world):
    """world-size 2 and 3, including H % world != 0 and a non-integer scale (unequal output strips)"""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, scale, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(42)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    full = oracle.sr_pipeline(img, luts_g, scale, scale)
    parts = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)]
    for r in range(world):
        assert np.load(tmp_path / ("ok_%d.npy" % r)).all()
        assert np.array_equal(np.load(tmp_path / ("whole_%d.npy" % r)), full)
    assert np.array_equal(np.concatenate(parts, axis=0), full)


def _grad_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import torch.nn as nn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lerf_pytorch_amd import dist as ldist

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.zeros(5, 3))
            self.b = nn.Parameter(torch.zeros(7))
            self.c = nn.Parameter(torch.zeros(2, 2))          # never receives a gradient on rank 1
    m = M()
    m.a.grad = torch.full((5, 3), float(rank + 1))
    m.b.grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    if rank == 0:
        m.c.grad = torch.ones(2, 2)
    ldist.allreduce_grads(m)
    q.put((rank, m.a.grad.clone().numpy(), m.b.grad.clone().numpy(), m.c.grad.clone().numpy()))
    dist.destroy_process_group()


def test_allreduce_grads_world2():
    """flat-bucket gradient averaging of the LUT fine-tuning path (gloo, world size 2)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, b, c in res:
        assert np.allclose(a, 1.5)
        assert np.allclose(b, np.arange(7) * 1.5)
        assert np.allclose(c, 0.5)
