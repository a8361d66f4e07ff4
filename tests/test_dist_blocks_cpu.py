"""CPU tests of the 2-D block partition (SURVEY.md 8e "2x4 blocks"): plan coverage, and gloo runs with world 4 (2 x 2)
and 8 (2 x 4) in which the halo exchange (edges + corners, one batch_isend_irecv) delivers exactly the pixels each rank
needs and the stitched per-block results reproduce the full frame.  The per-block compute here is the ORACLE (the
checker); the product compute path needs a GPU (tests/test_gpu_general.py::test_blocks_emulated_equal_full_frame)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lerf_pytorch_amd  # noqa: F401
from lerf_pytorch_amd import dist as ldist


@pytest.mark.parametrize("H,W,grid,S,scale", [(64, 96, (2, 4), 2, 2.0), (45, 50, (2, 2), 2, 1.5), (2160, 3840, (2, 4), 2, 2.0),
                                              (1080, 1920, (2, 4), 4, 2.0), (77, 131, (2, 3), 2, 2.4), (90, 61, (1, 3), 4, 3.0)])
def test_block_plan_covers_the_frame_and_its_support(oracle, H, W, grid, S, scale):
    lr, _, _, _ = oracle.sr_axis_tables(H, oracle.out_size(H, scale), scale, S)
    lc, _, _, _ = oracle.sr_axis_tables(W, oracle.out_size(W, scale), scale, S)
    owned = np.zeros((H, W), np.int32)
    outs = np.zeros((len(lr), len(lc)), np.int32)
    for r in range(grid[0] * grid[1]):
        p = ldist.BlockPlan(H, W, grid, r, S, lr, lc)
        assert p.halo == 3 + 3 + S // 2
        assert p.check_support(lr, lc)
        owned[p.y0:p.y1, p.x0:p.x1] += 1
        i0, i1, j0, j1 = p.out_rect()
        outs[i0:i1, j0:j1] += 1
        ry, rx, rh, rw = p.roi
        assert (rh, rw) == (p.y1 - p.y0, p.x1 - p.x0) and ry + rh <= p.local_hw[0] and rx + rw <= p.local_hw[1]
        if W % 4 == 0 and all((k * W // grid[1]) % 4 == 0 for k in range(grid[1] + 1)):
            assert p.local_hw[1] % 4 == 0            # the local row pitch keeps the aligned dword loads of interior tiles
    assert (owned == 1).all() and (outs == 1).all()


def test_block_grid_and_single_round_of_tiles(oracle):
    assert ldist.block_grid(8) == (2, 4) and ldist.block_grid(4) == (2, 2) and ldist.block_grid(2) == (1, 2)
    H, W = 2160, 3840
    lr, _, _, _ = oracle.sr_axis_tables(H, 2 * H, 2.0, 2)
    lc, _, _, _ = oracle.sr_axis_tables(W, 2 * W, 2.0, 2)
    for r in range(8):
        p = ldist.BlockPlan(H, W, (2, 4), r, 2, lr, lc)
        _, _, rh, rw = p.roi
        assert -(-rh // 64) * -(-rw // 64) == 255     # one round of workgroups on 256 CUs (strips: 300)


@pytest.mark.parametrize("H,W,grid", [(24, 24, (2, 4)), (40, 40, (1, 8)), (13, 200, (2, 2))])
def test_blocks_smaller_than_the_halo_are_refused(oracle, H, W, grid):
    """a block narrower / lower than the 7-pixel halo needs pixels of a NON-adjacent rank, which the
    8-neighbour exchange never delivers -- the plan must refuse instead of letting the kernel read uninitialised halo"""
    lr, _, _, _ = oracle.sr_axis_tables(H, 2 * H, 2.0, 2)
    lc, _, _, _ = oracle.sr_axis_tables(W, 2 * W, 2.0, 2)
    with pytest.raises(ValueError, match="smaller than the halo"):
        ldist.BlockPlan(H, W, grid, 0, 2, lr, lc)


def test_check_support_sees_uncovered_halo_pixels(oracle):
    """check_support verifies COVERAGE (owned block + what the adjacent blocks send), not just the local extent"""
    H, W, grid = 64, 64, (2, 2)
    lr, _, _, _ = oracle.sr_axis_tables(H, 2 * H, 2.0, 2)
    lc, _, _, _ = oracle.sr_axis_tables(W, 2 * W, 2.0, 2)
    p = ldist.BlockPlan(H, W, grid, 0, 2, lr, lc)
    assert p.check_support(lr, lc)
    q = ldist.BlockPlan(H, W, grid, 0, 2, lr, lc)
    q.neighbours = lambda: [n for n in ldist.BlockPlan.neighbours(q) if n[1] == 0 or n[2] == 0]      # drop the corner peer
    assert not q.check_support(lr, lc)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, grid, port, H, W, scale, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import lerf_oracle as O
    from conftest import ASSETS
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)        # same frame on every rank
    lr, _, _, _ = O.sr_axis_tables(H, O.out_size(H, scale), scale, 2)
    lc, _, _, _ = O.sr_axis_tables(W, O.out_size(W, scale), scale, 2)
    plan = ldist.BlockPlan(H, W, grid, rank, 2, lr, lc)
    buf = ldist.BlockBuffer(plan, 2, 3, torch.uint8, None, lr, lc)
    own = torch.from_numpy(img[plan.y0:plan.y1, plan.x0:plan.x1].copy())
    ok = True
    for rep in range(2):                                          # persistent buffer, reused across steps
        buf.ext.zero_()
        buf.own.copy_(torch.stack([own, own]))
        ext = buf.exchange()
        ok = ok and ext.data_ptr() == buf.ext.data_ptr()
        ok = ok and np.array_equal(ext[1].numpy(), img[plan.ylo:plan.yhi, plan.xlo:plan.xhi])
    # two parts of a batch, the exchange of part 2 posted before part 1 is consumed (dist.sr_batch_pipelined)
    parts = [ldist.BlockBuffer(plan, 1, 3, torch.uint8, None, lr, lc) for _ in range(2)]
    for k, b in enumerate(parts):
        b.ext.zero_()
        b.own.copy_(((own.to(torch.int32) + k) % 256).to(torch.uint8).unsqueeze(0))
    got = ldist.sr_batch_pipelined(None, parts, plan, None, compute=lambda e, o: e.clone())
    for k, g in enumerate(got):
        ok = ok and np.array_equal(g[0].numpy(), ((img[plan.ylo:plan.yhi, plan.xlo:plan.xhi].astype(np.int32) + k) % 256).astype(np.uint8))
    # per-block compute with the checker: LUT stages on the local frame alone (wrong only in the outer ring of an
    # artificial border, which stage 3 of the owned outputs never reads), stage 3 with the GLOBAL geometry
    luts = O.load_luts(os.path.join(ASSETS, "lerf-g"))
    feat_s, hq_s = O.lut_stages(ext[0].numpy(), luts, 3)
    feat = np.zeros((H, W, 3), np.uint8)
    hq = np.zeros((H, W, 3, 3), np.uint8)
    feat[plan.ylo:plan.yhi, plan.xlo:plan.xhi], hq[plan.ylo:plan.yhi, plan.xlo:plan.xhi] = feat_s, hq_s
    i0, i1, j0, j1 = plan.out_rect()
    mine = O.to_u8(O.resize_u8(feat, hq, scale, scale))[i0:i1, j0:j1]
    rects = [ldist.BlockPlan(H, W, grid, r, 2, lr, lc).out_rect() for r in range(world)]
    whole = ldist.gather_blocks(torch.from_numpy(mine.copy()), rects, (len(lr), len(lc)))
    np.save(os.path.join(tmp, "whole_%d.npy" % rank), whole.numpy())
    np.save(os.path.join(tmp, "ok_%d.npy" % rank), np.array([ok, plan.check_support(lr, lc)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,scale,grid", [(45, 50, 2.0, (2, 2)), (43, 90, 1.5, (2, 4)), (50, 49, 3.0, (2, 2))])
def test_gloo_block_halo_exchange_and_stitch(tmp_path, oracle, luts_g, H, W, scale, grid):
    """world 4 and 8 incl. H % 2 != 0, W % 4 != 0 and a non-integer scale (unequal output rectangles)"""
    world = grid[0] * grid[1]
    port = _free_port()
    mp.spawn(_worker, args=(world, grid, port, H, W, scale, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    full = oracle.sr_pipeline(img, luts_g, scale, scale)
    for r in range(world):
        assert np.load(tmp_path / ("ok_%d.npy" % r)).all()
        assert np.array_equal(np.load(tmp_path / ("whole_%d.npy" % r)), full)


@pytest.mark.parametrize("H,W,grid,S,scale", [(2160, 3840, (2, 4), 2, 2.0), (1080, 1920, (2, 2), 2, 1.5), (540, 700, (1, 3), 4, 3.0), (300, 300, (2, 2), 2, 2.0)])
def test_block_parts_partition_the_block(oracle, H, W, grid, S, scale):
    """dist.block_parts: interior + border rectangles tile the owned block, their output rectangles tile the block's output, the
    interior reads nothing outside the owned block (its launch can run under the halo exchange), small blocks stay whole"""
    lr = oracle.sr_axis_tables(H, oracle.out_size(H, scale), scale, S)[0]
    lc = oracle.sr_axis_tables(W, oracle.out_size(W, scale), scale, S)[0]
    for rank in range(grid[0] * grid[1]):
        plan = ldist.BlockPlan(H, W, grid, rank, S, lr, lc)
        parts = ldist.block_parts(plan, lr, lc, 64)
        own = np.zeros((plan.y1 - plan.y0, plan.x1 - plan.x0), int)
        out = np.zeros((plan.i1 - plan.i0, plan.j1 - plan.j0), int)
        for p in parts:
            ya, yb, xa, xb = p["rect"]
            ia, ib, ja, jb = p["out"]
            own[ya - plan.y0:yb - plan.y0, xa - plan.x0:xb - plan.x0] += 1
            out[ia - plan.i0:ib - plan.i0, ja - plan.j0:jb - plan.j0] += 1
            if p["interior"] and ib > ia and jb > ja:
                r12 = 6
                lo_r, hi_r = lr[ia:ib].min() - r12, lr[ia:ib].max() + S - 1 + r12
                lo_c, hi_c = lc[ja:jb].min() - r12, lc[ja:jb].max() + S - 1 + r12
                assert max(lo_r, 0) >= plan.y0 and min(hi_r, H - 1) < plan.y1 and max(lo_c, 0) >= plan.x0 and min(hi_c, W - 1) < plan.x1
        assert (own == 1).all() and (out == 1).all()
        assert sum(p["interior"] for p in parts) <= 1
