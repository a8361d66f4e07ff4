#!/usr/bin/env python3
"""Benchmark of the LeRF LUT resampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4,5}] [--mode frames|strips|blocks]
                    [--scale S] [--channels {1,3,4}] [--sustained SECONDS]

Default = BASELINE.json configs[1], the configuration the headline metric is quoted on: LeRF-G LUT x2 SR,
1920x1080 -> 3840x2160 RGB.  One "step" = one pass of the hot path (stage-1 LUTs -> stage-2 LUTs -> spatially
varying resampling, uint8 HWC in / uint8 HWC out) over a batch of `--frames` synthetic frames already resident in HBM.

--gpus N > 1: when not already running under torch.distributed.run (no RANK in the environment) this process
stays GPU-free and spawns N rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set,
one rank per GPU, RCCL); a failing rank makes the whole command fail.  Under torch.distributed.run it is one rank.
  --mode frames (default): every rank processes its own batch -- frames are independent, no data-path collective
                           (weak scaling; the barrier only brackets the timed region);
  --mode strips:           every frame is split into LR strips over the ranks with an RCCL halo exchange of raw
                           uint8 rows (strong scaling on the same batch);
  --mode blocks:           every frame is split into a 2-D grid of blocks (8 ranks: 2 x 4; edges + corners exchanged in
                           one RCCL group): a 2160x3840 frame is 255 tiles per rank, one round of workgroups.

--config: 1 = 256x256 tile through the CPU oracle port (plumbing; 1 thread and all cores) beside the GPU,
          2 = headline, 3 = LeRF-L x1.5/x2.0 (and x2/x2), 4 = LeRF-G homographic warp 1080p -> 4K (isc / osc
          matrices of SURVEY.md 8d), 5 = batch of 8 frames 2160x3840 -> 4320x7680 (strips by default when N > 1).

Prints ONE JSON line (rank 0).  `roofline` prices the launch against HBM bandwidth using the ALGORITHMIC bytes of
SURVEY.md 8(d); `cpu_baseline` is the C port of the oracle timed on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C = 3                                                   # channels per pixel (--channels overrides it for config 2)
LUT_BYTES = {"lerf-g": 1753941, "lerf-l": 751689}      # 3 x 83521 + 6 x 83521 x oC
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
M_ISC = [[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]]
M_OSC = [[4.1, 0.4, 30.0], [0.5, 3.8, 25.0], [8e-5, 1.2e-4, 1.0]]


# --------------------------------------------------------------------------------------------- synthetic inputs
def synth_frames(kind, n, seed, H, W, channels=None):
    """SURVEY.md 8(d) inputs.  'noise': uniform uint8 (worst case for LUT locality);
    'natural': low-pass field (box blur radius 8, 3 passes) + 5 % uniform noise.  channels: default = the run's (--channels)."""
    C = globals()["C"] if channels is None else int(channels)
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (n, H, W, C), dtype=np.uint8)
    out = np.empty((n, H, W, C), np.uint8)
    for i in range(n):
        x = rng.random((H, W, C))
        for _ in range(3):
            for ax in (0, 1):
                k = 17
                pad = [(0, 0)] * 3
                pad[ax] = (k // 2 + 1, k // 2)
                cs = np.cumsum(np.pad(x, pad, mode="reflect"), axis=ax)
                hi = [slice(None)] * 3
                lo = [slice(None)] * 3
                hi[ax] = slice(k, None)
                lo[ax] = slice(0, -k)
                x = (cs[tuple(hi)] - cs[tuple(lo)]) / k
        x = (x - x.min()) / (x.max() - x.min())
        x = x * 255.0 + (rng.random((H, W, C)) - 0.5) * 0.05 * 255.0
        out[i] = np.clip(np.round(x), 0, 255).astype(np.uint8)
    return out


def traffic_key(cfg, S, C, scale, frames, input_kind, variant=""):
    """key of a workload in profiles/hbm_traffic.json (tools/summarize_profile.py writes it, bench.py looks it up)"""
    return "config%d_S%d_C%d_x%gx%g_f%d_%s%s" % (cfg, S, C, scale[0], scale[1], frames, input_kind, ("_" + variant) if variant else "")


def kernel_source_sha():
    """sha256 over the HIP sources: ties recorded PMC numbers (profiles/hbm_traffic.json) to the kernels they were
    measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lerf-pytorch_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------------------------- CPU baseline legs
def host_cpu_budget():
    """(threads the baseline may use, why): min of the visible processors, the affinity mask and the cgroup CPU quota.
    The boxes of the pool show 256 processors to a container that is granted 16 (cpu.max): OpenMP threads beyond the
    quota are throttled, not run (profiles/r03_cpu_scaling.txt)."""
    n = os.cpu_count() or 1
    why = "os.cpu_count()"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "affinity mask"
    except AttributeError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, why = max(1, int(quota)), "cgroup CPU quota of this container (%d processors visible)" % (os.cpu_count() or 1)
    return n, why


def _oracle_luts(model):
    from oracle import lerf_oracle
    return lerf_oracle.load_luts(os.path.join(ROOT, "lerf-pytorch_amd", "assets", "models", model), linear=model == "lerf-l")


def cpu_baseline_sr(frames_u8, model, sh, sw, budget_s=12.0, threads=None, max_n=64, S=2):
    """C port of the oracle (oracle/lerf_oracle.c, OpenMP) on this host, bounded sample.  threads=1: one core."""
    from oracle import c_oracle
    luts = _oracle_luts(model)
    budget_thr, why = host_cpu_budget()
    c_oracle.set_threads(min(threads, budget_thr) if threads is not None else budget_thr)
    thr = c_oracle.threads()
    # one untimed call: the work area and the output frame are allocated and first-touched here (caller-owned scratch of
    # lerf_oracle_sr_u8_ws, cached by the wrapper), not inside the timed loop
    out = c_oracle.sr_u8(frames_u8[0], luts, sh, sw, S=S, linear=model == "lerf-l")
    t0 = time.perf_counter()
    n = 0
    while True:
        out = c_oracle.sr_u8(frames_u8[n % len(frames_u8)], luts, sh, sw, S=S, linear=model == "lerf-l", out=out)
        n += 1
        if time.perf_counter() - t0 >= budget_s or n >= max_n:
            break
    dt = time.perf_counter() - t0
    H, W = frames_u8.shape[1:3]
    return {
        "value": round(n * out.shape[0] * out.shape[1] / dt / 1e6, 4), "unit": "Mpix/s", "cores": thr, "kind": "port",
        "sample": "%d frame(s) %dx%d->%dx%d, same synthetic input, oracle/lerf_oracle.c (OpenMP, %d threads; host budget %d = %s), %.1f s"
                  % (n, W, H, out.shape[1], out.shape[0], thr, budget_thr, why, dt),
    }, out, n


def cpu_baseline_warp(frame_u8, matrix, out_hw, budget_s=12.0):
    from oracle import c_oracle
    luts = _oracle_luts("lerf-g")
    c_oracle.set_threads(host_cpu_budget()[0])
    thr = c_oracle.threads()
    t0 = time.perf_counter()
    n = 0
    while True:
        out, mask = c_oracle.warp_u8(frame_u8, luts, matrix, out_hw)
        n += 1
        if time.perf_counter() - t0 >= budget_s or n >= 32:
            break
    dt = time.perf_counter() - t0
    return {
        "value": round(n * out_hw[0] * out_hw[1] / dt / 1e6, 4), "unit": "Mpix/s", "cores": thr, "kind": "port",
        "sample": "%d frame(s) %dx%d->%dx%d warp + mask, same synthetic input, oracle/lerf_oracle.c (OpenMP, %d threads), %.1f s"
                  % (n, frame_u8.shape[1], frame_u8.shape[0], out_hw[1], out_hw[0], thr, dt),
    }, out, mask


def measure_lds_gather(torch, L, n_cu, target_ms=2.0):
    """ns of one CU's LDS per wave64 dword gather at (random, conflict-free) addresses: lerf_ubench_lds_gather timed with
    events on the current stream, sized to about `target_ms` per pattern."""
    lib = L._lib.lib()
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = []
    for pattern in (0, 1):
        def run(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            L._lib.check(lib.lerf_ubench_lds_gather(pattern, iters, n_cu, sink.data_ptr(), L._lib.current_stream()), "lerf_ubench_lds_gather")
            b.record()
            b.synchronize()
            return a.elapsed_time(b)
        run(50)
        probe = run(400)                                                        # ms for 400 iterations
        iters = int(max(400, min(200000, 400 * target_ms / max(probe, 1e-3))))
        ms = min(run(iters) for _ in range(3))
        res.append(ms * 1e6 / (iters * 10 * 16))                                # 16 waves x 10 gathers per iteration on each CU
    return res[0], res[1]


def end_to_end_legs(torch, L, eng, frame_u8, scale, B):
    """SURVEY 8(d) 'end-to-end incl. H2D/D2H, reported separately': host buffer in -> host buffer out for one 1080p frame,
    (a) pageable numpy through LerfEngine.sr, (b) pinned buffers with async copies on one stream, (c) stream.StreamingSR, B frames per launch:
    the kernel reads / writes the pinned host buffers itself (two slots), and (d) upload / launch / download of neighbouring batches on three
    streams (three slots).  Never `value`."""
    from lerf_pytorch_amd.stream import StreamingSR
    H, W = frame_u8.shape[:2]
    out = {}
    for _ in range(2):
        o = eng.sr(frame_u8, scale)
    torch.cuda.synchronize()
    n = 6
    t = time.perf_counter()
    for _ in range(n):
        o = eng.sr(frame_u8, scale)
    dt = (time.perf_counter() - t) / n
    opx = o.shape[0] * o.shape[1]
    out["pageable_numpy"] = {"ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(opx / dt / 1e6, 1)}
    pin_in = torch.from_numpy(frame_u8).pin_memory()
    pin_out = torch.empty(tuple(o.shape), dtype=torch.uint8).pin_memory()
    x = torch.empty(tuple(frame_u8.shape), dtype=torch.uint8, device="cuda")
    for k in range(n + 2):
        if k == 2:
            torch.cuda.synchronize()
            t = time.perf_counter()
        x.copy_(pin_in, non_blocking=True)
        y = eng.sr(x, scale)
        pin_out.copy_(y, non_blocking=True)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    out["pinned_async_copies"] = {"ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(opx / dt / 1e6, 1)}
    for transport, key in (("zero_copy", "streaming_zero_copy"), ("dma", "streaming_dma_pipeline")):
        st = StreamingSR(eng, (H, W), scale, frames_per_batch=B, transport=transport)
        for k in range(st.depth):
            st.input(k)[:] = frame_u8
        for _ in range(3):
            st.result(st.submit())
        nb = 9
        torch.cuda.synchronize()
        t = time.perf_counter()
        pend = []
        for _ in range(nb):
            if len(pend) == st.depth:
                st.result(pend.pop(0))
            pend.append(st.submit())
        while pend:
            st.result(pend.pop(0))
        dt = (time.perf_counter() - t) / (nb * B)
        out[key] = {"ms_per_frame": round(dt * 1e3, 3), "mpix_s": round(opx / dt / 1e6, 1), "frames_per_launch": B, "slots": st.depth}
        del st
    out["note"] = ("host uint8 frame in -> host uint8 frame out, %dx%d -> %dx%d, PCIe inside the figure (%.1f MB in + %.1f MB out per frame); "
                   "reported beside `value`, never as it" % (W, H, o.shape[1], o.shape[0], frame_u8.nbytes / 1e6, o.nbytes / 1e6))
    return out


def recorded_traffic(cfg, S, Cn, scale, frames, input_kind, sha, variant=""):
    """(bytes per launch, where from, the recorded entry) of profiles/hbm_traffic.json for a workload, only when it was measured on
    exactly these kernel sources; (None, why not, None) otherwise"""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    key = traffic_key(cfg, S, Cn, scale, frames, input_kind, variant)
    try:
        tj = json.load(open(tfile)).get("entries", {}).get(key)
    except Exception as e:
        return None, "none: %s" % e, None
    if tj is None:
        return None, "none: profiles/hbm_traffic.json has no entry for this workload (%s)" % key, None
    if tj.get("frames") == frames and tj.get("input") == input_kind and tj.get("kernel_src_sha16") == sha:
        return tj.get("bytes_per_launch"), "recorded, not measured in this run: %s (rocprofv3 --pmc passes of this workload on kernel sources sha %s)" % (
            tj.get("source"), sha), tj
    return None, "none: profiles/hbm_traffic.json was recorded for other kernel sources / workload (file sha %s, sources %s)" % (
        tj.get("kernel_src_sha16"), sha), None


def psnr_delta_block(torch, L):
    """The second half of BASELINE.json's metric ("...; PSNR delta vs ref"): the Set5 tables of the reference's scripts.sh:33-47
    (SR x2 / x3 / x4 by Y-PSNR with shave = scale, common/utils.py:138-151; homographic warps isc / osc by masked mPSNR,
    :168-175) through LerfEngine.sr_many / warp_many and the device metric kernels (lerf_metric_*), beside the reference's own
    per-image values (tests/golden/g5_set5.json, generated by importing the reference: tests/golden/gen_golden.py).  The
    images are the reference's test data (tests/data/Set5); nothing here reads /root/reference."""
    from PIL import Image
    from lerf_pytorch_amd import metrics
    data = os.path.join(ROOT, "tests", "data", "Set5")
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_set5.json")))
    names = ["baby", "bird", "butterfly", "head", "woman"]
    t0 = time.perf_counter()
    gts = {n: np.array(Image.open(os.path.join(data, "HR", n + ".png"))) for n in names}
    out = {"unit": "dB", "tables": {}, "max_abs_delta_db": 0.0,
           "published_scripts_sh": {"lerf-g": "35.71 32.02 30.15 | 33.81 27.89", "lerf-l": "34.84 30.72 29.13 | 32.90 27.13"}}
    worst = 0.0
    for model in ("lerf-g", "lerf-l"):
        eng = L.LerfEngine.shipped(model)
        jobs = [(s, n) for s in (2, 3, 4) for n in names]
        lrs = [np.array(Image.open(os.path.join(data, "LR_bicubic/rrLR_X%.2f_%.2f" % (s, s), n + ".png"))) for s, n in jobs]
        srs = eng.sr_many([eng._dev(a)[0] for a in lrs], [(float(s), float(s)) for s, _ in jobs])
        row = {}
        for s in (2, 3, 4):
            got = [metrics.psnr_y(gts[n], o, s) for (sj, n), o in zip(jobs, srs) if sj == s]
            want = [ref["sr"]["%s/x%d/%s" % (model, s, n)]["psnr_y"] for n in names]
            d = max(abs(g - w) for g, w in zip(got, want))
            worst = max(worst, d)
            row["sr_x%d" % s] = {"psnr_y": round(float(np.mean(got)), 4), "reference": round(float(np.mean(want)), 4), "max_abs_delta": round(d, 6)}
        for p in ("isc", "osc"):
            lw = [np.array(Image.open(os.path.join(data, p, n + ".png"))) for n in names]
            Ms = [np.array(ref["warp"]["%s/%s/%s" % (model, p, n)]["matrix"]) for n in names]
            ws = eng.warp_many([eng._dev(a)[0] for a in lw], Ms, [gts[n].shape[:2] for n in names])
            got = [metrics.mpsnr(o, gts[n], m) for n, (o, m) in zip(names, ws)]
            want = [ref["warp"]["%s/%s/%s" % (model, p, n)]["mpsnr"] for n in names]
            d = max(abs(g - w) for g, w in zip(got, want))
            worst = max(worst, d)
            row["warp_%s" % p] = {"mpsnr": round(float(np.mean(got)), 4), "reference": round(float(np.mean(want)), 4), "max_abs_delta": round(d, 6)}
        out["tables"][model] = row
    torch.cuda.synchronize()
    out["max_abs_delta_db"] = round(worst, 6)
    out["images"] = "Set5 (5 images) x {x2, x3, x4, isc, osc} x {lerf-g, lerf-l} = 50 outputs"
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


def other_config_legs(torch, L, ops, steps=20, warmup=5):
    """every BASELINE configuration on the driver's record.  Short legs beside the headline (never `value`): config 1
    (the 256 x 256 CPU-plumbing tile, product path beside one core of the C port), config 3
    (LeRF-L x1.5/x2.0), config 4 (LeRF-G warp, isc matrix), config 5 (4K -> 8K, 4 frames on one GPU) and config 2 at S = 4 --
    `warmup` + `steps` steps each, HIP events around every step, the product output of the last step compared with the C port of
    the oracle on one frame (bytes), and the workload's own recorded HBM traffic.  A failing leg reports its error and nothing else."""
    sha = kernel_source_sha()
    from oracle import c_oracle
    c_oracle.set_threads(host_cpu_budget()[0])
    legs = {}
    specs = [("config3_lerf_l_x1.5x2.0", 3, "lerf-l", 2, (1080, 1920), (1.5, 2.0), 8, "noise"),
             ("config4_warp_isc", 4, "lerf-g", 2, (1080, 1920), (2.0, 2.0), 8, "natural"),
             ("config5_4k_to_8k_one_gpu", 5, "lerf-g", 2, (2160, 3840), (2.0, 2.0), 4, "noise"),
             ("config2_support4", 2, "lerf-g", 4, (1080, 1920), (2.0, 2.0), 8, "noise")]
    # config 1 (BASELINE configs[0], the reference's own CPU-runnable case): one 256 x 256 RGB tile, x2 -- the C port on one core
    # beside the product path on the same tile, bytes compared (`bench.py --config 1` is the full line)
    t_leg = time.perf_counter()
    try:
        tile = np.random.default_rng(0).integers(0, 256, (256, 256, 3), dtype=np.uint8)
        luts1 = _oracle_luts("lerf-g")
        c_oracle.set_threads(1)
        c_oracle.sr_u8(tile, luts1, 2.0, 2.0, S=2)
        t0 = time.perf_counter()
        ref1 = c_oracle.sr_u8(tile, luts1, 2.0, 2.0, S=2)
        t_cpu = time.perf_counter() - t0
        c_oracle.set_threads(host_cpu_budget()[0])
        eng1 = L.LerfEngine.shipped("lerf-g", support=2, max_sigma=10.0)
        x1 = torch.from_numpy(tile).cuda()
        for _ in range(3):
            o1 = eng1.sr(x1, 2)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record()
            o1 = eng1.sr(x1, 2)
            b.record()
        torch.cuda.synchronize()
        kms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        d1 = o1.cpu().numpy() != ref1
        legs["config1_256x256_tile"] = {"mpix_s": round(512 * 512 / (kms * 1e-3) / 1e6, 2), "ms_per_step": round(kms, 4), "steps": 20, "warmup": 3,
                                        "frames_per_step": 1, "input": "noise", "workload": "lerf-g 256x256 -> 512x512, S=2, scale 2x2 (one launch pair, 16-row tiles)",
                                        "cpu_port_one_thread_mpix_s": round(512 * 512 / t_cpu / 1e6, 3),
                                        "parity_vs_cpu_port": {"mismatches": int(d1.sum()), "bytes": int(d1.size)},
                                        "leg_seconds": round(time.perf_counter() - t_leg, 2)}
        del eng1, x1, o1
    except Exception as e:
        legs["config1_256x256_tile"] = {"error": "%s: %s" % (type(e).__name__, e)}
    for name, cfg, model, S, (H, W), scale, B, input_kind in specs:
        t_leg = time.perf_counter()
        try:
            eng = L.LerfEngine.shipped(model, support=S, max_sigma=10.0)
            kind = "linear" if model == "lerf-l" else "gauss"
            if input_kind == "natural":
                two = synth_frames("natural", 2, 1000, H, W, channels=3)
                host = np.ascontiguousarray(np.tile(two, (B // 2 + 1, 1, 1, 1))[:B])
            else:
                host = synth_frames("noise", B, 1000, H, W, channels=3)
            frames = torch.from_numpy(host).cuda()
            ws = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(H, W, 3, B)), dtype=torch.uint8, device="cuda")
            if cfg == 4:
                oH, oW = 2 * H, 2 * W
                geo = ops.WarpGeometry((H, W), np.array(M_ISC), (oH, oW), eng.support)
                out = torch.empty((B, oH, oW, 3), dtype=torch.uint8, device="cuda")

                def step():
                    ops.warp_packed(ops.stages_packed(frames, eng.luts, workspace=ws), geo, kind, eng.max_sigma, out=out)
            else:
                geo = eng.sr_geometry((H, W), list(scale))
                oH, oW = geo.out_hw
                out = torch.empty((B, oH, oW, 3), dtype=torch.uint8, device="cuda")

                def step():
                    ops.sr_fused_u8(frames, eng.luts, geo, kind, eng.max_sigma, out=out, workspace=ws)
            for _ in range(warmup):
                step()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, b in ev:
                a.record()
                step()
                b.record()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            kms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            alg = B * (H * W * 3 + oH * oW * 3) + LUT_BYTES[model]
            traffic, tsrc, _ = recorded_traffic(cfg, S, 3, scale, B, input_kind, sha)
            leg = {"mpix_s": round(steps * B * oH * oW / dt / 1e6, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
                   "frames_per_step": B, "input": input_kind, "workload": "%s %dx%d -> %dx%d, S=%d%s" % (
                       model, W, H, oW, oH, S, ", homography M_ISC + validity mask" if cfg == 4 else ", scale %gx%g" % scale),
                   "roofline": {"bound": "hbm", "achieved": round(alg / (kms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                                "kernel_ms": round(kms, 4), "algorithmic_bytes_per_launch": alg}}
            # parity of the timed product output against the checker (one frame; the C port is test infrastructure)
            k = B - 1
            luts = _oracle_luts(model)
            if cfg == 4:
                cpu_out, cpu_mask = c_oracle.warp_u8(host[k], luts, np.array(M_ISC), (oH, oW))
                white = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
                white[4:H - 4, 4:W - 4] = 255
                mk = (ops.warp_hwc_u8(white, None, ops.WarpGeometry((H, W), np.array(M_ISC), (oH, oW), 1), "nearest", 1.0, out="f32") == 255).cpu().numpy()
                diff = out[k].cpu().numpy() != cpu_out
                leg["parity_vs_cpu_port"] = {"mismatches": int(diff.sum()), "mask_mismatches": int((mk != cpu_mask).sum()), "bytes": int(diff.size)}
            else:
                cpu_out = c_oracle.sr_u8(host[k], luts, scale[0], scale[1], S=S, linear=model == "lerf-l")
                diff = out[k].cpu().numpy() != cpu_out
                leg["parity_vs_cpu_port"] = {"mismatches": int(diff.sum()), "bytes": int(diff.size)}
            leg["leg_seconds"] = round(time.perf_counter() - t_leg, 2)
            legs[name] = leg
            del frames, out, ws, eng
        except Exception as e:                                  # a secondary leg must never cost the headline line
            legs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return legs


# --------------------------------------------------------------------------------------------- rank launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """GPU-free parent: start n copies of this command, one rank per GPU, and relay rank 0's JSON line.
    Nothing in this process has touched HIP (torch is not even imported), so no re-exec hazard exists."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    out0_file = tempfile.TemporaryFile()                        # rank 0's stdout (a pipe read would block on a hung rank)
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LERF_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    deadline = time.time() + 3600
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) or time.time() > deadline:
            for i, p in enumerate(procs):                       # a rank failed: stop exactly the children we started
                if rcs[i] is None:
                    p.kill()
                    rcs[i] = p.wait()
            break
        time.sleep(0.05)
    out0_file.seek(0)
    out0 = out0_file.read().decode()
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % bad, file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


# --------------------------------------------------------------------------------------------- main
def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4, 5], default=2, help="BASELINE.json configs[] (1-based)")
    ap.add_argument("--frames", type=int, default=8, help="frames per step (per GPU in frames mode; total batch in config 5)")
    ap.add_argument("--input", choices=["noise", "natural"], default=None)
    ap.add_argument("--support", type=int, choices=[2, 4], default=2, help="S (config 2 only: 4 = the class default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-input", action="store_true",
                    help="skip the short secondary runs (profiling: keeps the kernel trace to one workload)")
    ap.add_argument("--path", choices=["fused", "callsite", "classes-torch"], default="fused",
                    help="fused (default): the engine path the headline is quoted on; callsite: one 1080p frame through the UNCHANGED call "
                         "sites of eltr._worker (24 FourSimplexInterpFaster calls + set_shape + resize through the mirrors, host numpy in, "
                         "uint8 numpy out); classes-torch: the torch resampler twins on device tensors (training / validation shapes)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer-in / host-buffer-out legs (config 2)")
    ap.add_argument("--warp-fused", action="store_true", help="config 4: the tile-fused warp (lerf_warp_fused_u8: no packed stage outputs in HBM) instead of the three-launch path")
    ap.add_argument("--overlap-halo", action="store_true", help="blocks mode: launch the block's interior under the halo exchange, its border after it (dist.OverlappedBlock)")
    ap.add_argument("--no-psnr", action="store_true", help="skip the Set5 PSNR-delta-vs-reference block of the default line (config 2)")
    ap.add_argument("--unfused", action="store_true", help="config 2: time the 3-launch direct path instead")
    ap.add_argument("--scale", type=float, default=None, help="config 2: scale factor (default 2; > 4.9 takes the general kernels)")
    ap.add_argument("--channels", type=int, choices=[1, 3, 4], default=3, help="config 2: channels per pixel (1 / 4: general kernels)")
    ap.add_argument("--sustained", type=float, default=5.0,
                    help="seconds of back-to-back steps for the `sustained` leg (0 = skip; reported beside `value`, never replacing it)")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of liblerf_hip.so (A/B variants, tools/ab.sh); reported in the line")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs of the other BASELINE configurations (default run only)")
    ap.add_argument("--mode", choices=["frames", "strips", "blocks", "rows"], default=None,
                    help="frames: independent frames per GPU (default; config 5: the batch is divided over the GPUs); "
                         "blocks: every frame is split into a 2-D grid of blocks over the GPUs (8 GPUs: 2 x 4) with an RCCL halo "
                         "exchange (default for --config 5 with N > 1); strips: LR strips instead; rows (config 4): every "
                         "frame's OUTPUT rows over the GPUs, each rank warps from its band of the source (no exchange)")
    return ap.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus)                                  # never returns
    if args.config == 1:
        return run_config1(args)
    if args.path == "callsite":
        return run_callsite(args)
    if args.path == "classes-torch":
        return run_classes_torch(args)

    import torch
    import torch.distributed as dist
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    if args.lib:
        L._lib.use_library(args.lib)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # TEST HOOK (tests/test_gpu_fullsize.py): LERF_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and runs the barrier / the MAX over the
    # ranks on gloo, so the multi-rank code below can be exercised on a one-GPU box (RCCL refuses two ranks on one device).  The
    # line it prints says so ("backend": "gloo (ranks share one GPU: test hook)"); it is no measurement.
    share_gpu = world > 1 and os.environ.get("LERF_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    rccl_ranks = 1
    coll_dev = "cpu" if share_gpu else "cuda"
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
            rccl_ranks = 0
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the ranks RCCL itself reaches: an all-reduce of ones over xGMI, not the launcher's WORLD_SIZE.  A line whose
            # `ranks_reported_by_rccl` differs from --gpus is never printed: the run fails here instead.
            one = torch.ones(1, dtype=torch.int32, device="cuda")
            dist.all_reduce(one)
            rccl_ranks = int(one.item())
            if rccl_ranks != args.gpus:
                raise SystemExit("bench.py: --gpus %d but the RCCL all-reduce counted %d ranks" % (args.gpus, rccl_ranks))
    n_gpus = world
    if args.gpus != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    global C
    cfg = args.config
    model = "lerf-l" if cfg == 3 else "lerf-g"
    H, W = (2160, 3840) if cfg == 5 else (1080, 1920)
    scale = (1.5, 2.0) if cfg == 3 else (2.0, 2.0)
    if cfg == 2 and args.scale:
        scale = (float(args.scale), float(args.scale))
    if cfg == 2:
        C = args.channels
    elif args.channels != 3 or args.scale:
        raise SystemExit("--channels / --scale vary config 2 only")
    S = args.support if cfg == 2 else 2
    kind = "linear" if model == "lerf-l" else "gauss"
    input_kind = args.input or ("natural" if cfg == 4 else "noise")
    # config 5 over N > 1 GPUs: 2-D blocks by default (one round of tiles per rank; emulated per-rank time 97 % of the ideal for the
    # 8-frame batch against 84 % for strips, profiles/r05_8k_blocks.txt); --mode strips / frames choose otherwise
    mode = args.mode or ("blocks" if (cfg == 5 and world > 1) else "frames")
    strips = mode in ("strips", "blocks") and world > 1          # any partition of the FRAME over the ranks
    blocks = mode == "blocks" and world > 1
    if strips and cfg == 4:
        raise SystemExit("strips / blocks are SR partitions; the warp path scales by frames, or by output rows (--mode rows)")
    rows = mode == "rows" and world > 1 and cfg == 4             # every frame's OUTPUT rows divided over the ranks (dist.WarpRowPlan)
    if mode == "rows" and cfg != 4:
        raise SystemExit("--mode rows partitions the warp (config 4)")
    strips = strips or rows                                      # (strong scaling: one batch, every rank works on every frame)

    eng = L.LerfEngine.shipped(model, support=S, max_sigma=10.0)
    B = args.frames
    if cfg == 5 and not strips:
        if B % world:
            raise SystemExit("config 5, frames mode: the batch (%d) must divide over %d GPUs" % (B, world))
        B_local = B // world                                    # strong scaling: the same batch, divided
    else:
        B_local = B
    seed = 1000 + (0 if (cfg == 5) else rank)
    host = synth_frames(input_kind, B if cfg == 5 else B_local, seed, H, W)
    if cfg == 5 and not strips:
        host = host[rank * B_local:(rank + 1) * B_local]
    ms = eng.max_sigma
    buf_px = H * W

    def make_sr(scale_hw):
        nonlocal buf_px
        geo = eng.sr_geometry((H, W), list(scale_hw))
        oH, oW = geo.out_hw
        ws = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(H, W, C, B_local)), dtype=torch.uint8, device="cuda")
        if blocks:
            from lerf_pytorch_amd import dist as ldist
            lr_, lc_ = geo.host["left_r"], geo.host["left_c"]
            plan = ldist.BlockPlan(H, W, ldist.block_grid(world), rank, eng.support, lr_, lc_)
            lgeo = ldist.block_geometry(geo, plan)
            buf = ldist.BlockBuffer(plan, B_local, C, torch.uint8, torch.device("cuda"), lr_, lc_)
            buf.own.copy_(torch.from_numpy(np.ascontiguousarray(host[:, plan.y0:plan.y1, plan.x0:plan.x1])).cuda())
            out = ldist.block_output(plan, B_local, C, torch.device("cuda"))       # rows padded to 16 bytes (a view)

            buf_px = buf.ext.shape[1] * buf.ext.shape[2]

            lh, lw = plan.local_hw
            wsb = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(lh, lw, C, B_local)), dtype=torch.uint8, device="cuda")

            ovl = ldist.OverlappedBlock(eng, plan, geo) if args.overlap_halo else None

            def step(o=out):
                if ovl is not None:
                    # interior part launched under the halo transfers, border parts after them (dist.OverlappedBlock)
                    ovl.step(buf, o, workspace=wsb if B_local > 1 else False)
                    return
                ext = buf.exchange()
                # one frame per launch: ONE launch without the stage-1 pass (255 tiles fill the chip once); a batch: two launches
                # over the region of interest (stage 1 once per pixel over block + 4 px, round 4)
                ops.sr_fused_u8(ext, eng.luts, lgeo, kind, ms, out=o, workspace=wsb if B_local > 1 else False)
            return step, out, (oH, oW), None
        if strips:
            from lerf_pytorch_amd import dist as ldist
            plan = ldist.StripPlan(H, world, rank, eng.support, geo.host["left_r"])
            lgeo = geo.row_slice(plan.ylo, plan.yhi - plan.ylo, plan.i0, plan.i1)
            buf = ldist.StripBuffer(plan, B_local, W, C, torch.uint8, torch.device("cuda"))
            buf.own.copy_(torch.from_numpy(host[:, plan.y0:plan.y1]).cuda())      # this rank's rows of every frame
            out = torch.empty((B_local, plan.i1 - plan.i0, oW, C), dtype=torch.uint8, device="cuda")
            buf_px = buf.ext.shape[1] * buf.ext.shape[2]

            def step(o=out):
                ext = buf.exchange()
                ops.sr_fused_u8(ext, eng.luts, lgeo, kind, ms, out=o, workspace=ws)
            return step, out, (oH, oW), None
        frames = torch.from_numpy(host).cuda()
        out = torch.empty((B_local, oH, oW, C), dtype=torch.uint8, device="cuda")
        if args.unfused:
            def step(x=frames, o=out):
                for b in range(x.shape[0]):
                    feat, hq = ops.lut_stages(x[b], eng.luts)
                    o[b] = ops.resize_hwc_u8(feat, hq, geo, kind, ms, out="u8")
        else:
            def step(x=frames, o=out):
                ops.sr_fused_u8(x, eng.luts, geo, kind, ms, out=o, workspace=ws)
        return step, out, (oH, oW), frames

    def make_warp(matrix, out_hw):
        nonlocal buf_px
        if rows:
            # no exchange: a rank is handed its band of every frame (the source rows its output rows read + the LUT stages' reach)
            from lerf_pytorch_amd import dist as ldist
            plan = ldist.WarpRowPlan(H, W, np.array(matrix), out_hw, world, rank, eng.support)
            rgeo = plan.geometry()
            band = torch.from_numpy(np.ascontiguousarray(host[:, plan.b0:plan.b1])).cuda()
            outs = torch.empty((B_local, plan.i1 - plan.i0, out_hw[1], C), dtype=torch.uint8, device="cuda")
            wsr = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(plan.b1 - plan.b0, W, C, B_local)), dtype=torch.uint8, device="cuda")
            buf_px = (plan.b1 - plan.b0) * W

            def step(x=band, o=outs):
                ops.warp_packed(ops.stages_packed(x, eng.luts, workspace=wsr), rgeo, kind, ms, out=o)
            return step, outs, out_hw, band
        geo = ops.WarpGeometry((H, W), np.array(matrix), out_hw, eng.support)
        frames = torch.from_numpy(host).cuda()
        outs = torch.empty((B_local, out_hw[0], out_hw[1], C), dtype=torch.uint8, device="cuda")
        ws = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(H, W, C, B_local)), dtype=torch.uint8, device="cuda")

        fused = args.warp_fused and ops.warp_fused_supported(frames, eng.luts, geo, kind, ms)
        if fused:
            geo.tile_boxes(frames.device)                                          # (host pass over the output, once per homography)

        def step(x=frames, o=outs):
            if fused:
                ops.warp_fused_u8(x, eng.luts, geo, kind, ms, out=o, workspace=ws)   # s1_kernel + stage 2 / warp per source tile
                return
            packed = ops.stages_packed(x, eng.luts, workspace=ws)                  # one launch pair for the batch
            ops.warp_packed(packed, geo, kind, ms, out=o)                          # one launch for the batch (shared homography)
        return step, outs, out_hw, frames

    if cfg == 4:
        step, out, (oH, oW), frames = make_warp(M_ISC, (2 * H, 2 * W))
    else:
        step, out, (oH, oW), frames = make_sr(scale)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            fn()
            ev[k][1].record()
        barrier()
        dt = time.perf_counter() - t0
        launch_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return dt, launch_ms

    dt, launch_ms = timed(step, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    weak = cfg != 5 and not strips
    pix_per_step = (n_gpus if weak else 1) * (B_local if weak else B) * oH * oW
    value = pix_per_step * args.steps / dt / 1e6
    in_px_local = (buf_px if strips else H * W)                                    # LR pixels this rank reads per frame
    out_px_local = out.shape[1] * out.shape[2]                                     # output pixels this rank writes per frame
    alg_bytes = B_local * (in_px_local * C + out_px_local * C) + LUT_BYTES[model]  # this rank's launch
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9

    # sustained leg: seconds of back-to-back steps (clocks and temperatures settle; reported beside `value`, never instead)
    sustained = None
    if args.sustained > 0:
        barrier()
        t0 = time.perf_counter()
        n_sus = 0
        while True:
            for _ in range(20):
                step()
            n_sus += 20
            torch.cuda.synchronize()
            stop = torch.tensor([1.0 if time.perf_counter() - t0 >= args.sustained else 0.0], device=coll_dev)
            if world > 1:
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)
            if stop.item() > 0:
                break
        barrier()
        ds = time.perf_counter() - t0
        sustained = (n_sus, ds)

    extra = {}
    single = rank == 0 and world == 1 and not args.no_other_input
    if single and cfg in (2, 3, 5) and not args.unfused:
        # (a) the other input distribution, same shapes, short run
        other = "natural" if input_kind == "noise" else "noise"
        ho = synth_frames(other, 2, seed, H, W)
        xo = torch.from_numpy(np.ascontiguousarray(np.tile(ho, (B_local // 2 + 1, 1, 1, 1))[:B_local])).cuda()
        geo = eng.sr_geometry((H, W), list(scale))
        ws2 = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(H, W, C, B_local)), dtype=torch.uint8, device="cuda")
        o2 = torch.empty_like(out)
        d2, _ = timed(lambda: ops.sr_fused_u8(xo, eng.luts, geo, kind, ms, out=o2, workspace=ws2), max(5, args.steps // 2), 2)
        extra["mpix_s_other_input"] = {other: round(max(5, args.steps // 2) * B_local * oH * oW / d2 / 1e6, 2)}
        # (b) latency: ONE frame per launch (510 tiles on 256 CUs at 1080p: tile quantisation shows)
        o1 = torch.empty_like(out[:1])
        d1, l1 = timed(lambda: ops.sr_fused_u8(frames[:1], eng.luts, geo, kind, ms, out=o1, workspace=ws2), max(10, args.steps), 3)
        extra["latency_one_frame_per_launch"] = {"ms": round(l1, 4), "mpix_s": round(oH * oW / (l1 * 1e-3) / 1e6, 2)}
        del xo, o2, o1
    if single and cfg == 2 and not args.unfused and C == 3 and not args.no_end_to_end:
        try:
            extra["end_to_end"] = end_to_end_legs(torch, L, eng, host[0], list(scale), B_local)
        except Exception as e:                                  # a secondary leg must never cost the headline line
            extra["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if single and cfg == 2 and not args.unfused and C == 3 and not args.no_psnr:
        try:
            extra["psnr_delta_vs_ref_db"] = psnr_delta_block(torch, L)
        except Exception as e:
            extra["psnr_delta_vs_ref_db"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if single and cfg == 3:
        s2, o2, (oh2, ow2), _ = make_sr((2.0, 2.0))
        d2, _ = timed(s2, max(5, args.steps // 2), 2)
        extra["mpix_s_scale_2.0x2.0"] = round(max(5, args.steps // 2) * B_local * oh2 * ow2 / d2 / 1e6, 2)
        del o2
    if single and cfg == 4:
        s2, o2, _, _ = make_warp(M_OSC, (2 * H, 2 * W))
        d2, _ = timed(s2, max(5, args.steps // 2), 2)
        torch.cuda.synchronize()
        white = torch.zeros((H, W, C), dtype=torch.uint8, device="cuda")
        white[4:H - 4, 4:W - 4] = 255
        res_osc = {"mpix_s": round(max(5, args.steps // 2) * B_local * oH * oW / d2 / 1e6, 2)}
        for nm, M in (("isc", M_ISC), ("osc", M_OSC)):
            ngeo = ops.WarpGeometry((H, W), np.array(M), (oH, oW), 1)
            mk = ops.warp_hwc_u8(white, None, ngeo, "nearest", 1.0, out="f32") == 255
            frac = float(mk.float().mean().item())
            if nm == "isc":
                extra["valid_pixel_fraction_isc"] = round(frac, 4)
                extra["valid_mpix_s_isc"] = round(value * frac, 2)
            else:
                res_osc["valid_pixel_fraction"] = round(frac, 4)
                res_osc["valid_mpix_s"] = round(res_osc["mpix_s"] * frac, 2)
        extra["osc_matrix"] = res_osc
        del o2

    # recorded (not in-run) PMC numbers: only when they were measured on exactly these kernels and this workload
    traffic, tsrc, binding = None, None, None
    sha = kernel_source_sha()
    if not args.unfused and world == 1:
        traffic, tsrc, tj = recorded_traffic(cfg, S, C, scale, B_local, input_kind, sha, "warpfused" if (cfg == 4 and args.warp_fused) else "")
        if tj is not None:
            binding = {k: tj[k] for k in ("valu_instr_per_cu_cycle", "valu_busy", "lds_array_busy", "lds_bank_conflict_share",
                                          "l2_hit_rate", "kernel_trace_avg_us") if k in tj}

    chn = {1: "grey", 3: "RGB", 4: "RGBA"}[C]
    names = {2: ("Mpix/s LeRF-G x%g SR (2K->%s)" % (scale[0], "4K" if scale[0] == 2.0 else "%dx%d" % (oW, oH)),
                 "LeRF-G LUT x%g SR, 1920x1080->%dx%d %s uint8, S=%d, max_sigma=10 (BASELINE configs[1]%s)"
                 % (scale[0], oW, oH, chn, S, "" if (scale[0] == 2.0 and C == 3 and S == 2) else ", varied")),
             3: ("Mpix/s LeRF-L x1.5/x2.0 SR (2K input)", "LeRF-L LUT anisotropic SR x1.5/x2.0, 1920x1080->3840x1620 RGB uint8, S=2 (BASELINE configs[2])"),
             4: ("Mpix/s LeRF-G homographic warp (2K->4K)", "LeRF-G LUT homographic warp, 1920x1080->3840x2160 RGB uint8, isc-like matrix, S=2 (BASELINE configs[3])"),
             5: ("Mpix/s LeRF-G x2 SR (4K->8K, batch 8)", "LeRF-G LUT x2 SR, batch of %d frames 3840x2160->7680x4320 RGB uint8, S=2 (BASELINE configs[4])" % B)}
    if blocks:
        from lerf_pytorch_amd import dist as ldist
        par = "2-D blocks of every frame over %d GPUs (%d x %d grid), RCCL halo exchange of %d raw uint8 rows / columns with up to 8 neighbours (one batch_isend_irecv), one pack and one unpack launch" % (
            (n_gpus,) + ldist.block_grid(n_gpus) + (3 + 3 + S // 2,))
    elif rows:
        par = "output rows of every frame over %d GPUs (dist.WarpRowPlan): each rank holds its band of the source rows (+ 6 rows of LUT reach), no data-path collective" % n_gpus
    elif strips:
        par = "LR strips of every frame over %d GPUs, RCCL halo exchange of %d raw uint8 rows per side (batch_isend_irecv)" % (n_gpus, 3 + 3 + S // 2)
    elif cfg == 5:
        par = "the batch divided over %d GPU(s), whole frames, no data-path collective" % n_gpus
    else:
        par = "independent frames per GPU, no data-path collective"
    res = {
        "metric": names[cfg][0], "value": round(value, 2), "unit": "Mpix/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak" if weak else "strong",
        "vs_baseline": None, "dtype": "i32+f32 (u8 io)", "data": "synthetic",
        "config": {"workload": names[cfg][1], "baseline_config": cfg, "frames_per_step_per_gpu": B_local, "input": input_kind,
                   "path": "unfused-3-launch" if args.unfused else (("warp_fused_u8 (tile-fused: s1_kernel + stage 2 / warp per source tile)" if args.warp_fused
                                                                     else "stages_packed + warp_packed") if cfg == 4 else "sr_fused_u8"),
                   "mode": mode if strips else "frames", "parallelism": par, "ranks_reported_by_rccl": rccl_ranks,
                   "backend": ("gloo (ranks share one GPU: test hook, not a measurement)" if share_gpu else ("nccl" if world > 1 else None)),
                   "channels": C, "scale": list(scale),
                   "library": os.path.relpath(L._lib.LIB_PATH, ROOT)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                     "kernel_ms": round(launch_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                     "binding_resource": "lds_gather (the LUT gathers at the random-address rate of the LDS, see roofline_lds; then VALU issue. HBM is "
                                         "the nominal roofline only: %.2f B per output pixel; DESIGN.md section 5)"
                                         % ((alg_bytes - LUT_BYTES[model]) / (B_local * out_px_local)),
                     "binding_resources_from_pmc": binding},
        # LR (input) pixels per second: what the LUT stages see -- the fair comparison between scale factors
        "lr_mpix_s": round(value * (H * W) / (oH * oW), 2),
    }
    # The resource that binds (round 3): every LR pixel-channel costs 60 byte gathers (stage 1) + 60 gathers (stage 2: dword
    # entries for LeRF-G, bytes for LeRF-L) from LUTs staged in LDS, the stage-2 ones also on the tile halo; a wave64 gather at
    # random addresses takes 3.53 ns of a CU's LDS (32 lanes on 32 banks per cycle: 7.6 cycles, tools/ubench/lds_gather.hip,
    # profiles/r03_lds_gather_and_clocks.txt), conflict-free it would take 0.9.  achieved = wave-gathers per second and CU.
    if not args.unfused:
        r3 = 0 if cfg == 4 else S // 2              # the EMIT kernels of the warp path look up the tile itself, no stage-3 ring
        halo = ((64 + 2 * r3) * (192 // C + 2 * r3) * C) / float(64 * 192)
        wave_gathers = B_local * H * W * C * 60.0 * (1.0 + halo) / 64.0
        n_cu = int(torch.cuda.get_device_properties(dev_index).multi_processor_count)
        try:
            lds_ns, lds_ns_free = measure_lds_gather(torch, L, n_cu)           # measured in THIS run on THIS chip (~2 ms each)
        except Exception:                                                      # (never seen; the recorded figures of profiles/r04_bench.json then)
            lds_ns, lds_ns_free = 2.67, 1.01
        ach = wave_gathers / (launch_ms * 1e-3) / n_cu / 1e6
        res["roofline_lds"] = {"bound": "lds_gather", "achieved": round(ach, 2),
                               "peak": round(1e3 / lds_ns, 2), "unit": "M wave-gathers/s per CU",
                               "frac": round(ach * lds_ns / 1e3, 4),
                               "peak_conflict_free": round(1e3 / lds_ns_free, 2), "frac_conflict_free": round(ach * lds_ns_free / 1e3, 4),
                               "ns_per_wave_gather_random": round(lds_ns, 3), "ns_per_wave_gather_conflict_free": round(lds_ns_free, 3),
                               "compute_units": n_cu, "wave_gathers_per_launch": int(wave_gathers),
                               "note": "both peaks measured in this run by lerf_ubench_lds_gather (one 1024-thread workgroup per CU, ten ds_read_b32 per "
                                       "wave and iteration into a 134-KB table): `peak` = random addresses, what a data-dependent LUT gather can get "
                                       "(uniform noise spreads 32 lanes over 32 banks like a hash); `peak_conflict_free` = lane-linear addresses, "
                                       "2 LDS cycles per wave-instruction"}
        res["roofline"]["bound_note"] = "nominal (contract): the binding resource is the LDS gather rate, see roofline_lds"
    if binding and "valu_instr_per_cu_cycle" in binding:
        # the second resource (recorded PMC of these kernel sources): VALU wave-instructions issued per CU-cycle against the 2.0
        # a CU can issue in runs of simple instructions; the kernels' mixed streams get one instruction per 4-cycle pass and
        # SIMD = 1.0 per CU-cycle (profiles/r03_issue_rates.txt)
        res["roofline_valu"] = {"bound": "valu", "achieved": binding["valu_instr_per_cu_cycle"], "peak": 2.0,
                                "unit": "wave-instr/CU-cycle", "frac": round(binding["valu_instr_per_cu_cycle"] / 2.0, 4),
                                "source": tsrc}
        res["roofline_valu"]["note"] = "one instruction per 4-cycle pass and SIMD is what the kernels' mixed streams reach (1.0 per CU-cycle): profiles/r03_issue_rates.txt"
    if sustained:
        res["sustained"] = {"seconds": round(sustained[1], 2), "steps": sustained[0],
                            "value": round(pix_per_step * sustained[0] / sustained[1] / 1e6, 2), "unit": "Mpix/s",
                            "ms_per_step": round(sustained[1] / sustained[0] * 1e3, 4)}
    res.update(extra)

    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and not strips:
        if cfg == 4:
            cb, cpu_out, cpu_mask = cpu_baseline_warp(host[0], np.array(M_ISC), (oH, oW))
            res["cpu_baseline"] = cb
            step()
            torch.cuda.synchronize()
            white = torch.zeros((H, W, C), dtype=torch.uint8, device="cuda")
            white[4:H - 4, 4:W - 4] = 255
            mk = (ops.warp_hwc_u8(white, None, ops.WarpGeometry((H, W), np.array(M_ISC), (oH, oW), 1), "nearest", 1.0, out="f32") == 255).cpu().numpy()
            diff = np.abs(out[0].cpu().numpy().astype(int) - cpu_out.astype(int))
            res["parity_vs_cpu_port"] = {"max_abs_diff_u8": int(diff.max()), "mismatches": int((diff != 0).sum()),
                                         "mask_mismatches": int((mk != cpu_mask).sum())}
        else:
            budget = 12.0 if cfg != 5 else 20.0
            if cfg == 2 and S == 2:
                # scaling points of the port: one core on the 256x256 tile of BASELINE config 1 (a 1080p frame takes about a
                # minute on one core), 32 threads on the frame, then every hardware thread (the figure in `cpu_baseline`)
                tile = np.random.default_rng(0).integers(0, 256, (1, 256, 256, C), dtype=np.uint8)
                cb1, _, _ = cpu_baseline_sr(tile, model, scale[0], scale[1], budget_s=5.0, threads=1, max_n=8)
                res["cpu_baseline_single_thread"] = cb1
                half = max(1, host_cpu_budget()[0] // 2)
                cbh, _, _ = cpu_baseline_sr(host, model, scale[0], scale[1], budget_s=5.0, threads=half, S=S)
                res["cpu_baseline_half_budget"] = cbh
            cb, cpu_out, n_cpu = cpu_baseline_sr(host, model, scale[0], scale[1], budget_s=budget, S=S)
            res["cpu_baseline"] = cb
            # the OTHER baseline, quoted beside the port: the reference's own numpy path cannot run here (its files do not travel);
            # it was timed on this frame size in the 8-vCPU build container (692 s per 1080p frame)
            cb["reference_numpy_mpix_s"] = {"value": 0.0120, "unit": "Mpix/s", "where": "8-vCPU build container, single-threaded numpy, 692 s per 1080p -> 4K frame (SURVEY.md section 6)"}
            if cfg == 2 and S == 2:
                cb["scaling"] = {"threads_1_mpix_s": res["cpu_baseline_single_thread"]["value"],
                                 "threads_%d_mpix_s" % res["cpu_baseline_half_budget"]["cores"]: res["cpu_baseline_half_budget"]["value"],
                                 "threads_%d_mpix_s" % cb["cores"]: cb["value"],
                                 "speedup_over_one_thread": round(cb["value"] / max(res["cpu_baseline_single_thread"]["value"], 1e-9), 1),
                                 "note": "threads = the CPU budget of this container, see profiles/r03_cpu_scaling.txt"}
            # the timed product output must equal the checker's (<= 1 LSB)
            ref_idx = (n_cpu - 1) % len(host)
            step()
            torch.cuda.synchronize()
            diff = np.abs(out[ref_idx].cpu().numpy().astype(int) - cpu_out.astype(int))
            res["parity_vs_cpu_port"] = {"max_abs_diff_u8": int(diff.max()), "mismatches": int((diff != 0).sum())}
    default_run = cfg == 2 and S == 2 and C == 3 and scale == (2.0, 2.0) and not args.unfused and input_kind == "noise"
    if rank == 0 and world == 1 and default_run and not args.no_other_configs and not args.no_other_input:
        del out, frames
        torch.cuda.empty_cache()
        res["other_configs"] = other_config_legs(torch, L, ops)
    if rank == 0:
        # every mode, every N: the printed line's rank count is the launch's (share_gpu, the gloo test hook, reports 0 RCCL ranks)
        assert res["n_gpus"] == args.gpus and res["config"]["ranks_reported_by_rccl"] == (0 if share_gpu else args.gpus), res["config"]
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_callsite(args):
    """VERDICT round 3, missing #3: the path north_star literally promises -- `resample.model / eval_lut_sr / eval_lut_warp call
    sites are unchanged`.  tools/callsite_driver.py states the caller's protocol (resample/eval_lut_sr.py:541-665) against the
    mirrored names; three legs, one 1080p noise frame each: (naive) every call returns a numpy array = a device round trip per
    call, round 3's behaviour; (lazy) the mirrors return device-backed arrays (lerf_pytorch_amd.lazy) and the caller's numpy
    calls run in HBM; (lazy+asdevice) one added line uploads the frame first, so rot90 / pad of stage 1 run there as well."""
    import torch
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import lazy
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import callsite_driver as cd
    H, W = 1080, 1920
    frame = synth_frames(args.input or "noise", 1, 1000, H, W)[0]
    img = frame.astype(np.float32)
    from lerf_pytorch_amd import luts as lutmod
    luts = cd.float_luts(lutmod.load_lut_arrays(os.path.join(lutmod.ASSET_DIR, "lerf-g")))
    interp, pads, resizer = cd.mirror_api(linear=False, support=2, max_sigma=10)
    eng = L.LerfEngine.shipped("lerf-g", support=2, max_sigma=10.0)
    want = eng.sr(frame, 2)
    legs = {}

    # time spent INSIDE the library (the 24 + 2 mirrored calls and the final materialisation, host clock incl. the device waits they
    # contain) against the caller's own numpy between them: with a host image the caller's 12 np.rot90 + np.pad(edge) of stage 1
    # (25 ms each on a fresh 25-MB array) are its own and no drop-in can take them away
    inside = [0.0]

    def timed_call(f):
        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                inside[0] += time.perf_counter() - t
        return g

    class TimedResizer(object):
        def __init__(self, r):
            self.r = r
            self.set_shape = timed_call(r.set_shape)
            self.resize = timed_call(r.resize)

    interp_t, resizer_t = timed_call(interp), TimedResizer(resizer)

    def run(name, enabled, first, reps):
        lazy.set_enabled(enabled)
        try:
            out = None
            for k in range(reps + 1):
                if k == 1:
                    torch.cuda.synchronize()
                    inside[0] = 0.0
                    t0 = time.perf_counter()
                x = timed_call(lazy.asdevice)(img) if first else img
                res = cd.worker_sr(interp_t, pads, resizer_t, luts, x, (2.0, 2.0))
                out = timed_call(np.asarray)(res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        finally:
            lazy.set_enabled(True)
        legs[name] = {"ms_per_frame": round(dt * 1e3, 2), "mpix_s": round(out.shape[0] * out.shape[1] / dt / 1e6, 2),
                      "inside_library_ms": round(inside[0] / reps * 1e3, 2), "callers_own_numpy_ms": round((dt - inside[0] / reps) * 1e3, 2),
                      "bytes_equal_engine_path": bool(np.array_equal(out, want))}
        return dt, inside[0] / reps

    t_lazy, in_lazy = run("lazy_device_arrays", True, False, max(3, args.steps // 5))
    t_dev, _ = run("lazy_plus_asdevice_line", True, True, max(3, args.steps // 5))
    t_naive, in_naive = run("naive_numpy_round_trips", False, False, 2)
    t_eng = []
    for k in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sr(frame, 2)
        t_eng.append(time.perf_counter() - t0)
    legs["engine_sr_same_frame"] = {"ms_per_frame": round(min(t_eng) * 1e3, 2), "note": "LerfEngine.sr(uint8 numpy) -> uint8 numpy: the fast path of INTEGRATION.md (2 launches), pageable H2D / D2H inside"}
    res = {"metric": "Mpix/s LeRF-G x2 SR (2K->4K), unchanged call sites", "value": legs["lazy_device_arrays"]["mpix_s"], "unit": "Mpix/s",
           "n_gpus": 1, "steps": max(3, args.steps // 5), "warmup": 1, "ms_per_step": legs["lazy_device_arrays"]["ms_per_frame"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i16 numerators + f64 class API", "data": "synthetic",
           "config": {"workload": "one 1920x1080 RGB frame -> 3840x2160 through the reference's call sequence (resample/eval_lut_sr.py:541-665): "
                                  "24 FourSimplexInterpFaster calls + set_shape + resize + the caller's numpy calls; host float32 HWC in, host uint8 HWC out",
                      "baseline_config": 2, "path": "callsite (tools/callsite_driver.py against lerf_pytorch_amd mirrors)", "input": args.input or "noise"},
           "legs": legs, "speedup_lazy_over_naive": round(t_naive / t_lazy, 1),
           "speedup_inside_library_lazy_over_naive": round(in_naive / max(in_lazy, 1e-9), 1),
           "speedup_lazy_plus_one_line_over_naive": round(t_naive / t_dev, 1),
           "reference_numpy_same_frame_s": 692.0,
           "note": "the class-API kernels, not the tile-fused path: each FourSimplexInterpFaster call is one LUT pass as the caller asked for "
                   "it (24 launches of the LDS-resident kernel lut_interp_lds_kernel, 22 of them accumulating into the caller's sum), one "
                   "resampler launch (uint8 cell kernel when the results are deferred, float64 per-pixel kernel otherwise); reference numpy "
                   "on the build container: 692 s for this frame size (SURVEY.md section 6)"}
    print(json.dumps(res))


def run_classes_torch(args):
    """The torch twins (A9) on device tensors at the reference's own shapes: the training step [16,1,48,48] x4
    (train_model.py:348-349, option.py:18) and a validation frame [1,3,1080,1920] x2; set_shape once, resize timed."""
    import torch
    from lerf_pytorch_amd.resize_right import resize_right2d_torch as T
    out = {}
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, shape, scale, S in (("train_16x1x48x48_x4_S4", (16, 1, 48, 48), 4, 4), ("train_16x1x48x48_x4_S2", (16, 1, 48, 48), 4, 2),
                                  ("eval_1x3x1080x1920_x2_S2", (1, 3, 1080, 1920), 2, 2), ("eval_1x3x1080x1920_x2_S4", (1, 3, 1080, 1920), 2, 4)):
        x = (torch.rand(shape, generator=g) * 255).round().cuda()
        hs = [torch.rand(shape, generator=g).cuda() for _ in range(3)]
        r = T.SteeringGaussianResize2dTorch(support_sz=S, device=torch.device("cuda"), max_sigma=10)
        t0 = time.perf_counter()
        r.set_shape(list(shape), scale_factors=scale)
        t_shape = time.perf_counter() - t0
        for _ in range(3):
            y = r.resize(x, *hs)
        n = 50 if shape[-1] < 100 else 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(n):
            y = r.resize(x, *hs)
        ev[1].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        opx = y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]
        out[name] = {"ms_per_call": round(dt * 1e3, 4), "device_ms_per_call": round(ev[0].elapsed_time(ev[1]) / n, 4),
                     "out_mpixch_s": round(opx / dt / 1e6, 1), "set_shape_ms": round(t_shape * 1e3, 3), "out_shape": list(y.shape)}
    lin = T.AmplifiedLinearResize2dTorch(device=torch.device("cuda"))
    shape = (1, 3, 1080, 1920)
    x = (torch.rand(shape, generator=g) * 255).round().cuda()
    a = torch.rand(shape, generator=g).cuda()
    lin.set_shape(list(shape), scale_factors=[1.5, 2.0])
    for _ in range(3):
        y = lin.resize(x, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        y = lin.resize(x, a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out["eval_linear_1x3x1080x1920_x1.5x2.0"] = {"ms_per_call": round(dt * 1e3, 4), "out_mpixch_s": round(y.numel() / dt / 1e6, 1), "out_shape": list(y.shape)}
    k = "eval_1x3x1080x1920_x2_S2"
    res = {"metric": "Mpix/s SteeringGaussianResize2dTorch x2 (2K->4K), device tensors", "value": round(out[k]["out_mpixch_s"] / 3, 1), "unit": "Mpix/s",
           "n_gpus": 1, "steps": 20, "warmup": 3, "ms_per_step": out[k]["ms_per_call"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64 arithmetic, f32 io", "data": "synthetic",
           "config": {"workload": "resize_right2d_torch twins (A9) on device tensors, float32 [B,C,H,W] in / out, stage 3 only (direct resize_kernel)",
                      "baseline_config": 2, "path": "classes-torch"},
           "legs": out}
    print(json.dumps(res))


def run_config1(args):
    """BASELINE configs[0]: one 256x256 RGB tile, LeRF-G x2 -- the reference's own CPU-runnable case.  `value` is the C port
    of the oracle on all host cores; the single-thread figure and (when a GPU is present) the product path on the same tile
    with its parity against the port ride along.  The reference itself (numpy) takes 15.3 s = 0.017 Mpix/s on this tile
    (BASELINE.md section 3, 8-vCPU build container)."""
    tile = np.random.default_rng(0).integers(0, 256, (1, 256, 256, C), dtype=np.uint8)
    cb_all, out_all, _ = cpu_baseline_sr(tile, "lerf-g", 2.0, 2.0, budget_s=8.0, max_n=200)
    cb_one, out_one, _ = cpu_baseline_sr(tile, "lerf-g", 2.0, 2.0, budget_s=8.0, threads=1, max_n=8)
    res = {"metric": "Mpix/s LeRF-G x2 SR, one 256x256 RGB tile on CPU (plumbing)", "value": cb_all["value"], "unit": "Mpix/s",
           "n_gpus": 0, "steps": 1, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "i32+f64", "data": "synthetic",
           "config": {"workload": "LeRF-G LUT x2 SR, 256x256->512x512 RGB uint8 tile, S=2 (BASELINE configs[0])", "baseline_config": 1,
                      "input": "noise", "path": "oracle/lerf_oracle.c (CPU port of the oracle)"},
           "cpu_baseline": cb_all, "cpu_baseline_single_thread": cb_one,
           "reference_numpy_build_container": {"value": 0.0172, "unit": "Mpix/s", "source": "BASELINE.md section 3"},
           "port_threads_agree": bool(np.array_equal(out_all, out_one))}
    try:
        import torch
        if torch.cuda.is_available():
            import lerf_pytorch_amd as L
            eng = L.LerfEngine.shipped("lerf-g")
            x = torch.from_numpy(tile).cuda()
            o = eng.sr(x, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                o = eng.sr(x, 2)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            d = np.abs(o[0].cpu().numpy().astype(int) - out_all.astype(int))
            res["n_gpus"] = 1
            res["gpu_same_tile"] = {"ms_per_call": round(ms, 4), "mpix_s": round(512 * 512 / (ms * 1e-3) / 1e6, 2),
                                    "max_abs_diff_u8": int(d.max()), "mismatches": int((d != 0).sum())}
    except ImportError:
        pass
    print(json.dumps(res))


if __name__ == "__main__":
    main()
