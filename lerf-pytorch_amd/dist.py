"""Multi-GPU partitioning of the SR path: horizontal LR strips, one per rank,
with a halo exchange of raw uint8 input rows between neighbouring ranks
(`torch.distributed`; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  There is no reduction anywhere in this workload: each output
pixel depends on a bounded LR neighbourhood, so the only
exchange is 3 + 3 + S/2 rows per side and frame.

Independent frames need no communication at all -- `bench.py` scales that way
("weak"); strips serve one large frame whose latency matters.
"""
from __future__ import annotations

import numpy as np

STAGE1_RADIUS = 3      # modes c,t reach 3 px, rotated 4 ways (eval_lut_sr.py:63-81, 548-553)
STAGE2_RADIUS = 3


def halo_rows(support: int) -> int:
    """input halo per side: stage 1 + stage 2 + stage 3 radii (7 for S=2, 8 for S=4)."""
    return STAGE1_RADIUS + STAGE2_RADIUS + support // 2


class StripPlan:
    """Row partition of an H x W frame over `world` ranks for a given SR geometry.

    left_r: the global first-source-row table (SrGeometry.host['left_r'] or the oracle's
    sr_axis_tables()[0]); ownership rule = the fused kernel's tile rule: an output row belongs
    to the strip that contains left_r[i] + S/2.
    """

    def __init__(self, H: int, world: int, rank: int, support: int, left_r):
        if world < 1 or not (0 <= rank < world):
            raise ValueError("bad rank/world")
        if H < world:
            raise ValueError("fewer rows than ranks")
        self.H, self.world, self.rank, self.S = int(H), int(world), int(rank), int(support)
        self.halo = halo_rows(self.S)
        self.y0 = rank * H // world                 # owned LR rows [y0, y1)
        self.y1 = (rank + 1) * H // world
        self.ylo = max(self.y0 - self.halo, 0)      # rows held locally after the exchange
        self.yhi = min(self.y1 + self.halo, H)
        key = np.asarray(left_r, dtype=np.int64) + self.S // 2
        self.i0 = 0 if rank == 0 else int(np.searchsorted(key, self.y0, side="left"))
        self.i1 = len(key) if rank == world - 1 else int(np.searchsorted(key, self.y1, side="left"))
        self.need_top = self.y0 - self.ylo          # rows to receive from rank-1
        self.need_bot = self.yhi - self.y1          # rows to receive from rank+1

    def owned(self):
        return self.y0, self.y1

    def out_rows(self):
        return self.i0, self.i1

    def check_support(self, left_r):
        """every source row of every owned output row is held locally"""
        lr = np.asarray(left_r)[self.i0:self.i1]
        if len(lr) == 0:
            return True
        lo = lr.min() - (STAGE1_RADIUS + STAGE2_RADIUS)
        hi = lr.max() + self.S - 1 + (STAGE1_RADIUS + STAGE2_RADIUS)
        return max(lo, 0) >= self.ylo and min(hi, self.H - 1) < self.yhi


class StripBuffer:
    """Persistent device (or host, for the gloo tests) storage of one rank's strip INCLUDING its halo rows:
    `ext` [N, yhi-ylo, W, C] is what the fused kernel reads; `own` is the view of the rank's own rows [y0, y1)
    inside it -- the producer (H2D copy, decoder, previous stage) writes there, so the strip is never copied again.
    The halo rows travel through two small contiguous staging tensors per direction (RCCL / gloo point-to-point
    transfers need contiguous memory; a [N, halo] slice of `ext` is not)."""

    def __init__(self, plan: StripPlan, N, W, C, dtype=None, device=None):
        import torch
        self.plan = plan
        dtype = dtype if dtype is not None else torch.uint8
        self.ext = torch.empty((N, plan.yhi - plan.ylo, W, C), dtype=dtype, device=device)
        self.own = self.ext[:, plan.need_top:plan.need_top + (plan.y1 - plan.y0)]
        mk = lambda rows: torch.empty((N, rows, W, C), dtype=dtype, device=device)
        self.send_up, self.send_dn = mk(plan.halo), mk(plan.halo)
        self.recv_top, self.recv_bot = mk(plan.need_top), mk(plan.need_bot)

    def exchange(self, group=None):
        """Fill the halo rows of `ext` from the neighbouring ranks: one send/recv pair per neighbour inside one
        batch_isend_irecv (= one ncclGroupStart/End on RCCL).  Returns `ext`.  World of 1: no-op."""
        import torch.distributed as dist
        p = self.plan
        if p.world == 1:
            return self.ext
        if (p.H // p.world) < p.halo:
            raise ValueError("strips thinner than the halo (%d rows) are not supported" % p.halo)
        h = p.y1 - p.y0
        ops = []
        if p.rank > 0:                              # exchange with the strip above
            self.send_up.copy_(self.own[:, :p.halo])
            ops.append(dist.P2POp(dist.isend, self.send_up, p.rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, self.recv_top, p.rank - 1, group))
        if p.rank < p.world - 1:                    # exchange with the strip below
            self.send_dn.copy_(self.own[:, h - p.halo:])
            ops.append(dist.P2POp(dist.isend, self.send_dn, p.rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, self.recv_bot, p.rank + 1, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if p.need_top:
            self.ext[:, :p.need_top].copy_(self.recv_top)
        if p.need_bot:
            self.ext[:, p.need_top + h:].copy_(self.recv_bot)
        return self.ext


def exchange_halos(own_rows, plan: StripPlan, group=None, buffer: "StripBuffer | None" = None):
    """own_rows: uint8 tensor [y1-y0, W, C] (or [N, y1-y0, W, C]) of this rank's rows.
    Returns the tensor extended to rows [ylo, yhi) after one send/recv pair per neighbour.
    Convenience form of StripBuffer: the rows are copied once into a (given or fresh) buffer; callers that
    produce their rows directly into `StripBuffer.own` avoid even that copy.
    Works with any torch.distributed backend; a world of 1 is a no-op."""
    batched = own_rows.dim() == 4
    x = own_rows if batched else own_rows.unsqueeze(0)
    N, h, W, C = x.shape
    assert h == plan.y1 - plan.y0
    if plan.world == 1:
        return own_rows
    if buffer is None:
        buffer = StripBuffer(plan, N, W, C, x.dtype, x.device)
    buffer.own.copy_(x)
    out = buffer.exchange(group)
    return out if batched else out[0]


def sr_strip(engine, ext_rows, plan: StripPlan, geo, out=None, workspace=None):
    """This rank's output rows [i0, i1) from its extended strip (rows [ylo, yhi)).
    `geo` = the GLOBAL SrGeometry of the frame; its row tables are rebased to the strip."""
    from . import ops
    local = geo.row_slice(plan.ylo, plan.yhi - plan.ylo, plan.i0, plan.i1)
    return ops.sr_fused_u8(ext_rows, engine.luts, local, engine.kind, engine.max_sigma, out=out, workspace=workspace)


def gather_strips(out_rows, counts, group=None):
    """All ranks' output strips -> the whole frame on every rank.  Strips differ in height whenever H % world != 0
    or the scale is not an integer; every rank therefore contributes max(rows) rows (its own, zero-padded) to ONE
    all_gather_into_tensor (equal sizes: a single RCCL all-gather, and gloo accepts it), and the padding is dropped
    when the frame is assembled.  counts: [(i0, i1)] per rank."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    rows = [b - a for a, b in counts]
    m = max(rows)
    rest = tuple(out_rows.shape[1:])
    mine = torch.zeros((m,) + rest, dtype=out_rows.dtype, device=out_rows.device)
    mine[:out_rows.shape[0]].copy_(out_rows)
    allb = torch.empty((world * m,) + rest, dtype=out_rows.dtype, device=out_rows.device)
    dist.all_gather_into_tensor(allb, mine, group=group)
    if all(r == m for r in rows):
        return allb
    return torch.cat([allb[r * m:r * m + rows[r]] for r in range(world)], dim=0)


def sr_frame_strips(engine, own_rows, H, scale, group=None, gather=False):
    """SR of one H x W frame distributed over the process group by LR strips.
    own_rows: this rank's rows [y0,y1) (uint8 [y1-y0,W,C] on the GPU).  Returns this rank's
    output rows, or the whole frame on every rank if gather=True."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    W = own_rows.shape[-2]
    geo = engine.sr_geometry((H, W), scale)
    plan = StripPlan(H, world, rank, engine.support, geo.host["left_r"])
    ext = exchange_halos(own_rows, plan, group)
    out = sr_strip(engine, ext, plan, geo)
    if not gather or world == 1:
        return out
    counts = [StripPlan(H, world, r, engine.support, geo.host["left_r"]).out_rows() for r in range(world)]
    return gather_strips(out, counts, group)


# --------------------------------------------------------------------------- data-parallel LUT fine-tuning
def allreduce_grads(model, group=None):
    """Average the parameter gradients of a fine-tuning model over the ranks: the data-parallel counterpart of the
    reference's nn.DataParallel wrapper (train_model.py:355-357, gpuNum > 1), one process per GPU.

    The nine LUT gradients (3 x 83521 + 6 x 83521 x oC floats, about 7 MB for LeRF-G) travel as ONE flat bucket -- a
    single RCCL all-reduce per step; on the point-to-point xGMI ring that is bandwidth-bound and far cheaper than nine
    latency-bound small ones.  Parameters without a gradient contribute zeros so that every rank reduces the same
    layout.  No-op for world size 1."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
