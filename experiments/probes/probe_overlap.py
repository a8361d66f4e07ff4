#!/usr/bin/env python3
"""Round 5: can a small-footprint stage-3 kernel run BESIDE the LUT kernels of the next batch (kernel-level overlap of the VALU-only
and the LDS-bound phases, which the persistent kernel of round 4 could not get inside one workgroup)?  Feasibility on config 4, whose
stage 3 already is a separate kernel: stream A runs stages_packed (s1_kernel + EMIT kernel) of batch k, stream B the packed warp of
batch k - 1; events order the two and guard the double-buffered packed maps.  Serial = everything on one stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = L.LerfEngine.shipped("lerf-g")
x = torch.from_numpy(bench.synth_frames("natural", 2, 1000, 1080, 1920)).cuda().repeat(N // 2, 1, 1, 1).contiguous()
geo = ops.WarpGeometry((1080, 1920), np.array(bench.M_ISC), (2160, 3840), 2)
outs = [torch.empty((N, 2160, 3840, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
ws = torch.empty(L._lib.lib().lerf_sr_fused_workspace_bytes(1080, 1920, 3, N), dtype=torch.uint8, device="cuda")


def serial(n):
    for _ in range(n):
        p = ops.stages_packed(x, eng.luts, workspace=ws)
        ops.warp_packed(p, geo, "gauss", 10.0, out=outs[0])


A, B = torch.cuda.Stream(), torch.cuda.Stream()
bufs = [torch.empty((N, 1080, 1920, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
lib = L._lib.lib()


def stages_into(buf):
    L._lib.check(lib.lerf_stages_packed_u8(x.data_ptr(), x.stride(0), N, 1080, 1920, 3, eng.luts.ref(), buf.data_ptr(), buf.stride(0),
                                           ws.data_ptr(), ws.numel(), L._lib.current_stream()), "stages")


def pipelined(n):
    done_s = [None, None]          # stages event per buffer
    done_w = [None, None]          # warp event per buffer (buffer free again)
    for k in range(n + 1):
        b = k & 1
        if k < n:
            with torch.cuda.stream(A):
                if done_w[b] is not None:
                    A.wait_event(done_w[b])
                stages_into(bufs[b])
                done_s[b] = torch.cuda.Event(); done_s[b].record(A)
        if k >= 1:
            pb = (k - 1) & 1
            with torch.cuda.stream(B):
                B.wait_event(done_s[pb])
                ops.warp_packed(bufs[pb], geo, "gauss", 10.0, out=outs[pb])
                done_w[pb] = torch.cuda.Event(); done_w[pb].record(B)
    torch.cuda.current_stream().wait_stream(A)
    torch.cuda.current_stream().wait_stream(B)


def T(f, n=30):
    f(3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    f(n)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


for rep in range(3):
    ts, tp = T(serial), T(pipelined)
    print("%d frames per batch: serial %.3f ms per batch (%.1f Gpix/s), two streams %.3f ms (%.1f Gpix/s): x%.3f" % (
        N, ts, N * 2160 * 3840 / ts / 1e6, tp, N * 2160 * 3840 / tp / 1e6, ts / tp))
# same bytes
serial(1); torch.cuda.synchronize(); ref = outs[0].clone()
pipelined(2); torch.cuda.synchronize()
print("bytes equal:", bool(torch.equal(outs[0], ref)), bool(torch.equal(outs[1], ref)))
