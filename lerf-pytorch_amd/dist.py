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
        if world > 1 and H // world < self.halo:
            # a strip's halo rows must all belong to its two NEIGHBOURS: thinner strips would need rows of ranks further away
            raise ValueError("strips thinner than the halo (%d rows) are not supported: %d rows over %d ranks" % (self.halo, H, world))
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
        self.N, self.W, self.C = int(N), int(W), int(C)
        self.ext = torch.empty((N, plan.yhi - plan.ylo, W, C), dtype=dtype, device=device)
        self.own = self.ext[:, plan.need_top:plan.need_top + (plan.y1 - plan.y0)]
        # one contiguous staging tensor per direction of travel, a segment per neighbour (RCCL / gloo point-to-point
        # transfers need contiguous memory; a [N, halo] slice of `ext` is not)
        seg = lambda rows: N * rows * W * C
        h = plan.y1 - plan.y0
        self.sends, self.recvs = [], []          # (peer, (y, x, h, w) in `ext`, byte offset, bytes)
        so = ro = 0
        if plan.rank > 0:
            self.sends.append((plan.rank - 1, (plan.need_top, 0, plan.halo, W), so, seg(plan.halo)))
            so += seg(plan.halo)
            if plan.need_top:
                self.recvs.append((plan.rank - 1, (0, 0, plan.need_top, W), ro, seg(plan.need_top)))
                ro += seg(plan.need_top)
        if plan.rank < plan.world - 1:
            self.sends.append((plan.rank + 1, (plan.need_top + h - plan.halo, 0, plan.halo, W), so, seg(plan.halo)))
            so += seg(plan.halo)
            if plan.need_bot:
                self.recvs.append((plan.rank + 1, (plan.need_top + h, 0, plan.need_bot, W), ro, seg(plan.need_bot)))
                ro += seg(plan.need_bot)
        self.send_buf = torch.empty(max(so, 1), dtype=dtype, device=device)
        self.recv_buf = torch.empty(max(ro, 1), dtype=dtype, device=device)

    def _copy(self, rects, staging, to_staging):
        if not rects:
            return
        if self.ext.is_cuda:                     # ONE launch for all rectangles (lerf_rect_copy_u8)
            from . import ops
            ops.rect_copy(self.ext, staging, [(y, x, h, w, off) for _, (y, x, h, w), off, _ in rects], to_staging)
            return
        for _, (y, x, h, w), off, nb in rects:   # host tensors (gloo tests): plain slicing
            seg = staging[off:off + nb].view(self.N, h, w, self.C)
            if to_staging:
                seg.copy_(self.ext[:, y:y + h, x:x + w])
            else:
                self.ext[:, y:y + h, x:x + w].copy_(seg)

    def exchange(self, group=None):
        """Fill the halo rows of `ext` from the neighbouring ranks: one pack launch, one send/recv pair per neighbour
        inside one batch_isend_irecv (= one ncclGroupStart/End on RCCL), one unpack launch.  Returns `ext`.  World of 1: no-op."""
        import torch.distributed as dist
        p = self.plan
        if p.world == 1:
            return self.ext
        if (p.H // p.world) < p.halo:
            raise ValueError("strips thinner than the halo (%d rows) are not supported" % p.halo)
        self.finish(self.post(group))
        return self.ext

    def post(self, group=None):
        """pack + post the transfers of this buffer; returns the work handles for finish()"""
        import torch.distributed as dist
        self._copy(self.sends, self.send_buf, True)
        ops_ = []
        for peer, _, off, nb in self.sends:
            ops_.append(dist.P2POp(dist.isend, self.send_buf[off:off + nb], peer, group))
        for peer, _, off, nb in self.recvs:
            ops_.append(dist.P2POp(dist.irecv, self.recv_buf[off:off + nb], peer, group))
        return dist.batch_isend_irecv(ops_) if ops_ else []

    def finish(self, works):
        """wait for the transfers of post() (RCCL: the compute stream waits, not the host) and unpack the halo"""
        for w in works:
            w.wait()
        self._copy(self.recvs, self.recv_buf, False)
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


# --------------------------------------------------------------------------- 2-D block partition ("2x4 blocks")
def block_grid(world: int):
    """(rows, cols) of the block grid for `world` ranks: as square as possible with cols >= rows (8 -> 2 x 4, 4 -> 2 x 2,
    6 -> 2 x 3, primes -> 1 x world)."""
    gy = int(np.floor(np.sqrt(world)))
    while world % gy:
        gy -= 1
    return gy, world // gy


class BlockPlan:
    """2-D partition of an H x W frame over a gy x gx grid of ranks (rank = ry * gx + rx) for a given SR geometry.

    Why blocks when strips exist: a 2160x3840 frame in 8 strips is 270 + 14 rows x 3840 columns per rank = 5 x 60 = 300
    tiles of 64 x 64 on 256 CUs (1.17 rounds of workgroups); in 2 x 4 blocks a rank owns 1080 x 960 LR pixels = 17 x 15 =
    255 tiles -- ONE round -- because the tile-fused kernel lays its tiles over the OWNED block only (lerf_sr_geo_t roi)
    and the halo pixels serve as tile halos.  Price: up to 8 neighbours (4 edges + 4 corners) in the halo exchange, still
    one batch_isend_irecv (one RCCL group).

    Ownership = the fused kernel's tile rule on both axes: output row i belongs to the block row that contains
    left_r[i] + S/2, output column j to the block column that contains left_c[j] + S/2.  The halo is 3 + 3 + S/2 pixels;
    the local column range is widened to multiples of `align` pixels (default 4) so that the local row pitch stays a
    4-byte multiple and interior tiles keep their aligned dword loads -- a wider halo is harmless.
    """

    def __init__(self, H, W, grid, rank, support, left_r, left_c, align=4):
        gy, gx = int(grid[0]), int(grid[1])
        world = gy * gx
        if not (0 <= rank < world):
            raise ValueError("bad rank/grid")
        if H < gy or W < gx:
            raise ValueError("fewer rows / columns than blocks")
        self.H, self.W, self.grid, self.world, self.rank, self.S = int(H), int(W), (gy, gx), world, int(rank), int(support)
        self.ry, self.rx = rank // gx, rank % gx
        self.halo = halo_rows(self.S)
        # the halo of a block must lie inside its 8 ADJACENT blocks (BlockBuffer exchanges with those only): every block at
        # least `halo` pixels high (if the grid has more than one row) and wide (more than one column).  Smaller blocks would
        # need pixels of ranks further away, which nobody sends -- the kernel would read uninitialised halo bytes.
        if (gy > 1 and H // gy < self.halo) or (gx > 1 and W // gx < self.halo):
            raise ValueError("blocks smaller than the halo (%d pixels) are not supported: %d x %d pixels over a %d x %d grid"
                             % (self.halo, H, W, gy, gx))
        self.y0, self.y1 = self.ry * H // gy, (self.ry + 1) * H // gy            # owned LR rows / columns
        self.x0, self.x1 = self.rx * W // gx, (self.rx + 1) * W // gx
        self.ylo, self.yhi = max(self.y0 - self.halo, 0), min(self.y1 + self.halo, H)
        a = max(int(align), 1)
        self.xlo = max((self.x0 - self.halo) // a * a, 0)
        self.xhi = min(-((-(self.x1 + self.halo)) // a) * a, W)
        kr = np.asarray(left_r, dtype=np.int64) + self.S // 2
        kc = np.asarray(left_c, dtype=np.int64) + self.S // 2
        self.i0 = 0 if self.ry == 0 else int(np.searchsorted(kr, self.y0, side="left"))
        self.i1 = len(kr) if self.ry == gy - 1 else int(np.searchsorted(kr, self.y1, side="left"))
        self.j0 = 0 if self.rx == 0 else int(np.searchsorted(kc, self.x0, side="left"))
        self.j1 = len(kc) if self.rx == gx - 1 else int(np.searchsorted(kc, self.x1, side="left"))

    # local frame = rows [ylo, yhi) x columns [xlo, xhi); the owned block sits at `roi` inside it
    @property
    def local_hw(self):
        return self.yhi - self.ylo, self.xhi - self.xlo

    @property
    def roi(self):
        return self.y0 - self.ylo, self.x0 - self.xlo, self.y1 - self.y0, self.x1 - self.x0

    def out_rect(self):
        return self.i0, self.i1, self.j0, self.j1

    def neighbours(self):
        """[(rank, dy, dx)] of the up to 8 blocks this one exchanges halos with."""
        gy, gx = self.grid
        out = []
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ny, nx = self.ry + dy, self.rx + dx
                if (dy or dx) and 0 <= ny < gy and 0 <= nx < gx:
                    out.append((ny * gx + nx, dy, dx))
        return out

    def _overlap(self, other: "BlockPlan"):
        """global rectangle of `other`'s OWNED block that lies inside this rank's local frame: (y, x, h, w) or None"""
        ya, yb = max(self.ylo, other.y0), min(self.yhi, other.y1)
        xa, xb = max(self.xlo, other.x0), min(self.xhi, other.x1)
        if ya >= yb or xa >= xb:
            return None
        return ya, xa, yb - ya, xb - xa

    def needed_rect(self, left_r, left_c):
        """global rectangle (y, x, h, w) of the source pixels the owned output pixels depend on, or None"""
        lr = np.asarray(left_r)[self.i0:self.i1]
        lc = np.asarray(left_c)[self.j0:self.j1]
        if len(lr) == 0 or len(lc) == 0:
            return None
        r12 = STAGE1_RADIUS + STAGE2_RADIUS
        ya, yb = max(int(lr.min()) - r12, 0), min(int(lr.max()) + self.S - 1 + r12, self.H - 1) + 1
        xa, xb = max(int(lc.min()) - r12, 0), min(int(lc.max()) + self.S - 1 + r12, self.W - 1) + 1
        return ya, xa, yb - ya, xb - xa

    def check_support(self, left_r, left_c):
        """every source pixel of every owned output pixel is held locally AND filled: it belongs to the owned block or to
        the part of an adjacent block that the halo exchange delivers (pixels of blocks further away are never sent)"""
        need = self.needed_rect(left_r, left_c)
        if need is None:
            return True
        ya, xa, h, w = need
        if ya < self.ylo or xa < self.xlo or ya + h > self.yhi or xa + w > self.xhi:
            return False
        have = np.zeros((h, w), dtype=bool)

        def mark(y0, x0, y1, x1):
            y0, x0, y1, x1 = max(y0, ya), max(x0, xa), min(y1, ya + h), min(x1, xa + w)
            if y0 < y1 and x0 < x1:
                have[y0 - ya:y1 - ya, x0 - xa:x1 - xa] = True

        mark(self.y0, self.x0, self.y1, self.x1)
        for peer, _, _ in self.neighbours():
            other = BlockPlan(self.H, self.W, self.grid, peer, self.S, np.zeros(1), np.zeros(1))
            inc = self._overlap(other)
            if inc is not None:
                mark(inc[0], inc[1], inc[0] + inc[2], inc[1] + inc[3])
        return bool(have.all())


class BlockBuffer:
    """Persistent storage of one rank's block INCLUDING its halo: `ext` [N, yhi-ylo, xhi-xlo, C] is what the fused kernel
    reads; `own` is the view of the rank's own block inside it (the producer writes there).  Per step: ONE pack launch
    gathers the up to 8 outgoing rectangles into one staging tensor (lerf_rect_copy_u8; torch slicing on CPU tensors for the
    gloo tests), one batch_isend_irecv moves a contiguous segment per neighbour, ONE unpack launch scatters the received
    segments into the halo of `ext`."""

    def __init__(self, plan: BlockPlan, N, C, dtype=None, device=None, left_r=None, left_c=None):
        import torch
        self.plan, self.N, self.C = plan, int(N), int(C)
        dtype = dtype if dtype is not None else torch.uint8
        lh, lw = plan.local_hw
        self.ext = torch.empty((N, lh, lw, C), dtype=dtype, device=device)
        ry, rx, rh, rw = plan.roi
        self.own = self.ext[:, ry:ry + rh, rx:rx + rw]
        # rectangles, in LOCAL coordinates of the sender / the receiver; both sides derive them from the same two plans
        self.sends, self.recvs = [], []          # (peer, (y, x, h, w), byte offset, bytes)
        so = ro = 0
        for peer, _, _ in plan.neighbours():
            other = BlockPlan(plan.H, plan.W, plan.grid, peer, plan.S, left_r if left_r is not None else np.zeros(1),
                              left_c if left_c is not None else np.zeros(1))
            out = other._overlap(plan)           # part of MY block the peer holds as halo
            if out is not None:
                y, x, h, w = out
                nb = N * h * w * C
                self.sends.append((peer, (y - plan.ylo, x - plan.xlo, h, w), so, nb))
                so += nb
            inc = plan._overlap(other)           # part of the PEER's block I hold as halo
            if inc is not None:
                y, x, h, w = inc
                nb = N * h * w * C
                self.recvs.append((peer, (y - plan.ylo, x - plan.xlo, h, w), ro, nb))
                ro += nb
        self.send_buf = torch.empty(max(so, 1), dtype=dtype, device=device)
        self.recv_buf = torch.empty(max(ro, 1), dtype=dtype, device=device)

    def _copy(self, rects, staging, to_staging):
        if not rects:
            return
        if self.ext.is_cuda:
            from . import ops
            ops.rect_copy(self.ext, staging, [(y, x, h, w, off) for _, (y, x, h, w), off, _ in rects], to_staging)
            return
        for _, (y, x, h, w), off, nb in rects:   # host tensors (gloo tests): plain slicing
            seg = staging[off:off + nb].view(self.N, h, w, self.C)
            if to_staging:
                seg.copy_(self.ext[:, y:y + h, x:x + w])
            else:
                self.ext[:, y:y + h, x:x + w].copy_(seg)

    def exchange(self, group=None):
        """Fill the halo of `ext` from the neighbouring blocks.  Returns `ext`.  World of 1: no-op."""
        import torch.distributed as dist
        if self.plan.world == 1:
            return self.ext
        self.finish(self.post(group))
        return self.ext

    def post(self, group=None):
        """pack + post the transfers of this buffer; returns the work handles for finish()"""
        import torch.distributed as dist
        self._copy(self.sends, self.send_buf, True)
        ops_ = []
        for peer, _, off, nb in self.sends:
            ops_.append(dist.P2POp(dist.isend, self.send_buf[off:off + nb], peer, group))
        for peer, _, off, nb in self.recvs:
            ops_.append(dist.P2POp(dist.irecv, self.recv_buf[off:off + nb], peer, group))
        return dist.batch_isend_irecv(ops_) if ops_ else []

    def finish(self, works):
        """wait for the transfers of post() (RCCL: the compute stream waits, not the host) and unpack the halo"""
        for w in works:
            w.wait()
        self._copy(self.recvs, self.recv_buf, False)
        return self.ext


def block_geometry(geo, plan: BlockPlan):
    """the GLOBAL SrGeometry rebased to the rank's local frame, tiles laid over the owned block"""
    lh, lw = plan.local_hw
    return geo.block_slice(plan.ylo, lh, plan.i0, plan.i1, plan.xlo, lw, plan.j0, plan.j1, roi=plan.roi)


def sr_block(engine, ext, plan: BlockPlan, geo, out=None, workspace="auto"):
    """This rank's output rectangle [i0, i1) x [j0, j1) from its extended block(s) [N, lh, lw, C].
    workspace="auto": ONE frame -> one launch (stage 1 recomputed on the tile halos: a block is sized to fill the chip once,
    a second launch would only add its tail); a BATCH of frames -> the two-launch path over the region of interest
    (stage 1 once per pixel over the owned block widened by 3 + S/2 pixels, then stages 2+3 over the owned block):
    8 frames of a 1080 x 960 block are 8 x 272 + 8 x 255 workgroups instead of 8 x 255 with 27 % more stage-1 work each.
    workspace=False / None / a tensor: as ops.sr_fused_u8."""
    from . import ops
    if isinstance(workspace, str):
        workspace = None if (ext.dim() == 4 and ext.shape[0] > 1) else False
    if out is None:
        out = block_output(plan, ext.shape[0] if ext.dim() == 4 else 1, ext.shape[-1], ext.device)
        if ext.dim() == 3:
            out = out[0]
    return ops.sr_fused_u8(ext, engine.luts, block_geometry(geo, plan), engine.kind, engine.max_sigma, out=out, workspace=workspace)


def block_parts(plan: BlockPlan, left_r, left_c, border=64):
    """The owned block cut into an INTERIOR rectangle and up to four BORDER rectangles (top / bottom strips over the block's
    whole width, left / right strips between them), `border` pixels thick (one tile) on the sides that have a neighbour.
    Nothing the interior's outputs read lies outside the owned block (border >= halo): its launch does not depend on the halo
    exchange and runs under it; the border rectangles follow once the halo is in.  Each part: dict(name, rect = (ya, yb, xa, xb)
    global LR pixels, out = (ia, ib, ja, jb) global output pixels by the block rule -- support centre inside the rectangle).
    The parts' output rectangles partition the block's own."""
    gy, gx = plan.grid
    if border < plan.halo:
        raise ValueError("the border must cover the halo (%d pixels)" % plan.halo)
    top = border if plan.ry > 0 else 0
    bot = border if plan.ry < gy - 1 else 0
    lft = border if plan.rx > 0 else 0
    rgt = border if plan.rx < gx - 1 else 0
    if plan.y1 - plan.y0 <= top + bot or plan.x1 - plan.x0 <= lft + rgt:
        return [dict(name="block", rect=(plan.y0, plan.y1, plan.x0, plan.x1), out=(plan.i0, plan.i1, plan.j0, plan.j1), interior=False)]
    ya, yb, xa, xb = plan.y0 + top, plan.y1 - bot, plan.x0 + lft, plan.x1 - rgt
    kr = np.asarray(left_r, dtype=np.int64) + plan.S // 2
    kc = np.asarray(left_c, dtype=np.int64) + plan.S // 2

    def rows(a, b):
        return (plan.i0 if a == plan.y0 else int(np.searchsorted(kr, a, side="left")),
                plan.i1 if b == plan.y1 else int(np.searchsorted(kr, b, side="left")))

    def cols(a, b):
        return (plan.j0 if a == plan.x0 else int(np.searchsorted(kc, a, side="left")),
                plan.j1 if b == plan.x1 else int(np.searchsorted(kc, b, side="left")))
    parts = [dict(name="interior", rect=(ya, yb, xa, xb), out=rows(ya, yb) + cols(xa, xb), interior=True)]
    for name, r in (("top", (plan.y0, ya, plan.x0, plan.x1)), ("bottom", (yb, plan.y1, plan.x0, plan.x1)),
                    ("left", (ya, yb, plan.x0, xa)), ("right", (ya, yb, xb, plan.x1))):
        if r[0] < r[1] and r[2] < r[3]:
            parts.append(dict(name=name, rect=r, out=rows(r[0], r[1]) + cols(r[2], r[3]), interior=False))
    return parts


def part_geometry(geo, plan: BlockPlan, part):
    """the GLOBAL SrGeometry rebased to the rank's local frame, tiles laid over one part of the owned block"""
    lh, lw = plan.local_hw
    ya, yb, xa, xb = part["rect"]
    ia, ib, ja, jb = part["out"]
    return geo.block_slice(plan.ylo, lh, ia, ib, plan.xlo, lw, ja, jb, roi=(ya - plan.ylo, xa - plan.xlo, yb - ya, xb - xa))


class OverlappedBlock:
    """One part (frame or batch) per step with the halo exchange UNDER the interior's kernels: post() the transfers, launch the
    interior part (reads the owned block only), finish() the transfers, launch the border parts.  The geometry slices are built
    once.  Same bytes as sr_block on the whole block (tests/test_gpu_fullsize.py)."""

    def __init__(self, engine, plan: BlockPlan, geo, border=64):
        self.engine, self.plan = engine, plan
        self.parts = block_parts(plan, geo.host["left_r"], geo.host["left_c"], border)
        self.geos = [part_geometry(geo, plan, p) for p in self.parts]

    def _launch(self, k, ext, out, workspace):
        from . import ops
        p = self.parts[k]
        ia, ib, ja, jb = p["out"]
        if ib <= ia or jb <= ja:
            return
        view = out[..., ia - self.plan.i0:ib - self.plan.i0, ja - self.plan.j0:jb - self.plan.j0, :]
        ops.sr_fused_u8(ext, self.engine.luts, self.geos[k], self.engine.kind, self.engine.max_sigma, out=view, workspace=workspace)

    def step(self, buf: "BlockBuffer", out, group=None, workspace="auto"):
        """buf.own holds this step's pixels; out: block_output(plan, ...).  Returns out."""
        ext = buf.ext
        if isinstance(workspace, str):
            workspace = None if ext.shape[0] > 1 else False
        works = buf.post(group) if self.plan.world > 1 else None
        first = [k for k, p in enumerate(self.parts) if p["interior"]]
        for k in first:
            self._launch(k, ext, out, workspace)
        if works is not None:
            buf.finish(works)
        for k in range(len(self.parts)):
            if k not in first:
                self._launch(k, ext, out, workspace)
        return out


def block_output(plan: BlockPlan, N, C, device):
    """Output tensor [N, i1-i0, j1-j0, C] of a rank's block as a VIEW of rows padded to a multiple of 16 bytes.  The blocks at
    the frame's left / right edge own 1919 / 1921 output columns at x2 (the half-pixel shift of the grid): dense rows of 5757
    bytes start at every phase of a dword, and the kernel then stores row by row, byte by byte (+13 % per launch, measured);
    the pitched rows keep its 16-byte store path (lerf_sr_geo_t.out_row_pitch)."""
    import torch
    h, w = plan.i1 - plan.i0, plan.j1 - plan.j0
    wp = -(-(w * C) // 16) * 16
    if wp % C:                                     # the padded pitch must still be a whole number of pixels for a [.., w, C] view
        wp = -(-(w * C) // (16 * C)) * (16 * C)
    buf = torch.empty((N, h, wp // C, C), dtype=torch.uint8, device=device)
    return buf[:, :, :w]


def sr_batch_pipelined(engine, buffers, plan, geo, outs=None, group=None, compute=None):
    """Halo exchange of part k + 1 of a batch under the kernels of part k.  `buffers`: BlockBuffer / StripBuffer objects
    (one per part of the batch, e.g. two halves of 4 frames), already filled through `.own`.  All parts are packed and their
    transfers posted first (RCCL runs them on its own stream; `wait()` only makes the compute stream wait for one part), then
    every part is unpacked and computed as soon as ITS halo has arrived.  Returns the list of outputs.
    compute(ext, out) defaults to sr_block / sr_strip on `plan`."""
    import torch.distributed as dist
    if compute is None:
        if isinstance(plan, BlockPlan):
            compute = lambda ext, out: sr_block(engine, ext, plan, geo, out=out)
        else:
            compute = lambda ext, out: sr_strip(engine, ext, plan, geo, out=out)
    outs = list(outs) if outs is not None else [None] * len(buffers)
    pending = []
    for b in buffers:
        pending.append(b.post(group) if plan.world > 1 else None)
    res = []
    for b, works, o in zip(buffers, pending, outs):
        if works is not None:
            b.finish(works)
        res.append(compute(b.ext, o))
    return res


def gather_blocks(out_block, rects, out_hw, group=None):
    """All ranks' output rectangles -> the whole frame on every rank: every rank contributes a (max rows x max cols) tile
    (its own, zero-padded) to ONE all_gather_into_tensor, the frame is assembled from the valid parts.
    out_block: [h, w, C] (or [N, h, w, C]); rects: [(i0, i1, j0, j1)] per rank."""
    import torch
    import torch.distributed as dist
    batched = out_block.dim() == 4
    x = out_block if batched else out_block.unsqueeze(0)
    world = len(rects)
    mh = max(r[1] - r[0] for r in rects)
    mw = max(r[3] - r[2] for r in rects)
    N, _, _, C = x.shape
    mine = torch.zeros((N, mh, mw, C), dtype=x.dtype, device=x.device)
    mine[:, :x.shape[1], :x.shape[2]].copy_(x)
    allb = torch.empty((world, N, mh, mw, C), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(allb.view(world * N, mh, mw, C), mine, group=group)
    frame = torch.empty((N, out_hw[0], out_hw[1], C), dtype=x.dtype, device=x.device)
    for r, (i0, i1, j0, j1) in enumerate(rects):
        frame[:, i0:i1, j0:j1].copy_(allb[r, :, :i1 - i0, :j1 - j0])
    return frame if batched else frame[0]


def sr_frame_blocks(engine, own_block, H, W, scale, grid=None, group=None, gather=False):
    """SR of one H x W frame distributed over the process group by 2-D blocks (default grid: block_grid(world)).
    own_block: this rank's pixels [y0:y1, x0:x1] (uint8 [h,w,C] on the GPU).  Returns this rank's output rectangle, or the
    whole frame on every rank if gather=True."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    grid = grid or block_grid(world)
    geo = engine.sr_geometry((H, W), scale)
    lr, lc = geo.host["left_r"], geo.host["left_c"]
    plan = BlockPlan(H, W, grid, rank, engine.support, lr, lc)
    if not plan.check_support(lr, lc):
        raise ValueError("the halo exchange of this block grid does not cover the pixels rank %d needs" % rank)
    buf = BlockBuffer(plan, 1, own_block.shape[-1], own_block.dtype, own_block.device, lr, lc)
    buf.own.copy_(own_block.unsqueeze(0))
    out = sr_block(engine, buf.exchange(group), plan, geo)[0]
    if not gather or world == 1:
        return out
    rects = [BlockPlan(H, W, grid, r, engine.support, lr, lc).out_rect() for r in range(world)]
    return gather_blocks(out, rects, geo.out_hw, group)


# --------------------------------------------------------------------------- data-parallel LUT fine-tuning
# --------------------------------------------------------------------------- homographic warp over ranks
class WarpRowPlan:
    """Partition of a homographic warp (resize_right/resize_right2d_numpy.py:306-407) over `world` ranks by OUTPUT rows: rank r owns
    the output rows [i0, i1) (all columns) and needs the SOURCE rows [b0, b1): the rows its taps read, found by projecting the
    rectangle's border pixels through inv(M) with the reference's own float64 steps (a linear-fractional map is monotone along a
    line: the extremes of the projected row coordinate lie on the rectangle's border), widened by the reach of the two LUT stages
    (3 + 3 rows, recomputed in the band like the halo of the SR partitions) -- clipped projections fall on the frame's border rows,
    which are then part of the band.  Nothing is exchanged between ranks: every rank is handed (or reads) its band of the input;
    outputs stay sharded or are gathered (gather_strips).  The bands of neighbouring ranks overlap."""

    def __init__(self, H, W, matrix, out_hw, world, rank, support=2):
        if world < 1 or not (0 <= rank < world):
            raise ValueError("bad rank/world")
        self.H, self.W, self.world, self.rank, self.S = int(H), int(W), int(world), int(rank), int(support)
        self.out_hw = (int(out_hw[0]), int(out_hw[1]))
        oH, oW = self.out_hw
        if oH < world:
            raise ValueError("fewer output rows than ranks")
        self.matrix = np.asarray(matrix, dtype=np.float64)
        self.i0, self.i1 = rank * oH // world, (rank + 1) * oH // world
        lo, hi = self._tap_rows(self.i0, self.i1)
        reach = STAGE1_RADIUS + STAGE2_RADIUS
        self.t0, self.t1 = lo, hi + 1                                  # rows the warp's taps read
        self.b0, self.b1 = max(lo - reach, 0), min(hi + 1 + reach, self.H)

    def _tap_rows(self, i0, i1):
        """(first, last) source row read by the taps of output rows [i0, i1): the border pixels of the rectangle, every one"""
        from . import _lib
        oH, oW = self.out_hw
        minv = np.linalg.inv(self.matrix)                              # :327
        pads = _lib.warp_pads(minv, (self.H, self.W), self.out_hw, self.S)
        ii = np.concatenate([np.full(oW, i0), np.full(oW, i1 - 1), np.arange(i0, i1), np.arange(i0, i1)]).astype(np.float64)
        jj = np.concatenate([np.arange(oW), np.arange(oW), np.zeros(i1 - i0), np.full(i1 - i0, oW - 1)]).astype(np.float64)
        Y = minv[1, 0] * jj + minv[1, 1] * ii + minv[1, 2]
        Wh = minv[2, 0] * jj + minv[2, 1] * ii + minv[2, 2]
        if not ((Wh > 0).all() or (Wh < 0).all()):
            return 0, self.H - 1                                       # the horizon crosses the rectangle: the whole frame
        gr = np.clip(Y / Wh, 0, self.H)                                # :338-339
        eps = float(np.finfo(np.float32).eps)
        lr = np.ceil(gr - self.S / 2 - eps).astype(np.int64) + pads[0]
        lo = np.clip(np.clip(lr, 0, self.H - 1) - pads[0], 0, self.H - 1)
        hi = np.clip(np.clip(lr + self.S - 1, 0, self.H - 1) - pads[0], 0, self.H - 1)
        return int(lo.min()), int(hi.max())

    def band(self):
        return self.b0, self.b1

    def out_rows(self):
        return self.i0, self.i1

    def geometry(self, support=None):
        """ops.WarpGeometry of this rank's output rows reading a band that starts at source row b0"""
        from . import ops
        oH, oW = self.out_hw
        return ops.WarpGeometry((self.H, self.W), self.matrix, self.out_hw, self.S if support is None else support,
                                out_rect=(self.i0, self.i1, 0, oW), src_y0=self.b0)


def warp_rows(engine, band_u8, plan: WarpRowPlan, border=4):
    """This rank's output rows [i0, i1) of the warp and their validity mask from its band of the input: uint8 [b1 - b0, W, C]
    (or a batch [N, ...] sharing the homography) -> (uint8 [.., i1 - i0, oW, C], bool mask [i1 - i0, oW, C]).  The LUT stages run
    on the band as a frame of its own (their values are wrong only within 6 rows of an artificial band edge, which no tap reads);
    the warp and the nearest-warp mask (resample/eval_lut_warp.py:197-204, 229) run with the WHOLE frame's geometry."""
    import torch
    from . import ops
    squeeze = band_u8.dim() == 3
    x = band_u8.unsqueeze(0) if squeeze else band_u8
    if x.shape[1] != plan.b1 - plan.b0 or x.shape[2] != plan.W:
        raise ValueError("the band must hold source rows [%d, %d)" % (plan.b0, plan.b1))
    geo = plan.geometry()
    packed = ops.stages_packed(x, engine.luts)
    out = ops.warp_packed(packed, geo, engine.kind, engine.max_sigma, out="u8")
    Cn = x.shape[-1]
    white = torch.zeros((plan.b1 - plan.b0, plan.W, Cn), dtype=torch.uint8, device=x.device)
    ya, yb = max(border, plan.b0), min(plan.H - border, plan.b1)
    if yb > ya:
        white[ya - plan.b0:yb - plan.b0, border:plan.W - border] = 255
    mask = ops.warp_hwc_u8(white, None, plan.geometry(support=1), "nearest", 1.0, out="f32") == 255
    return (out[0] if squeeze else out), mask


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
