// Persistent stages-2+3 kernel of the headline configuration (LeRF-G, 2x2 support, RGB, integer x2, two-launch path):
// included by lerf_fused_impl.h inside its namespace (it uses that file's helpers, Dims and the addtid macros).
//
// Why (round 4): in sr_fused_kernel the two binding resources are used one after the other -- the
// piece copies are LDS-WRITE-bound with the VALU idle (29 k cycles per tile), the lookups LDS-gather / VALU co-bound, and
// stage 3 VALU-bound with the LDS idle (41 k).  A diagnostic build that dealt the instruction mix of a tile's stage 3 out
// over the tile's 18 piece-copy sub-phases ran the step in 2.09 ms against 2.30 (profiles/r04_experiments.txt): under the
// piece stores of the other waves that VALU work is almost free.  This kernel does it for real:
//
//   * one workgroup per CU walks tiles b, b + G, b + 2 G, ... of the XCD-ordered tile sequence (G = grid size, a multiple
//     of 8: a workgroup stays on its XCD's contiguous part of the sequence);
//   * stages 2 of tile N run exactly as in sr_fused_kernel (binning, 18 phases of piece store -> lookups); the finalised
//     (hq0, hq1, hq2, feat) dwords of the tile go to a 52-KB scratch of the workgroup in global memory (it lives in L2)
//     instead of LDS;
//   * stage 3 of tile N is DEFERRED: its 2 x 2-output block tasks are cut into per-channel sub-tasks (131 VALU instructions
//     each, the arithmetic of run_blocks operation for operation) and executed during tile N+1's phases, one or two per
//     phase, right behind the wave's own piece stores and in front of the phase barrier -- where the LDS is busy with the
//     other waves' stores, the wave's 36 prefetch registers are dead and the VALU has nothing else to do.  Taps come from the
//     scratch (dword loads, L1 / L2 hits), the geometry of an exact x2 grid is eight distance constants and two origins
//     per tile (read from the tables once per tile), the output bytes are stored straight from registers;
//   * rounding ties (lerf_stage3.h) are queued in LDS and re-evaluated in float64 behind the tile's last sub-task, as before;
//   * the last tile of a workgroup is drained without overlap.
//
// Everything else -- LDS use (feat tile + piece + tie queue), the slot loop, the piece transfer -- is sr_fused_kernel's.
// Same bytes: the deferred sub-tasks run the same float32 operations in the same order on the same taps.

static_assert(NBIN == 3 || NBIN == 4, "bins");
#ifndef LERF_PERSIST_ORDER
#define LERF_PERSIST_ORDER 3
#endif

struct PersistDims {
    using D = Dims<2, false, false>;
    static constexpr int OFF_TQ = D::OFF_X + PIECE_LDS;
    static constexpr int TQ_CAP = 2048;
    static constexpr int OFF_CTL = OFF_TQ + TQ_CAP * 4;
    static constexpr int LDS_BYTES = OFF_CTL + 512;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(D::OFF_TAB + NW * 4 * 4 <= D::OFF_X + PIECE_LDS, "binning scratch under the piece");
    static constexpr int SCR_DWORDS = (D::NH + 255) / 256 * 256;      // per-workgroup scratch: the tile's hyper region as dwords
};

// what the deferred stage 3 of a tile needs to know about it: one block of the LDS control area, written once per tile.
// The hot part is five dwords (one ds_read_b128 + one ds_read_b32 per sub-task); the exact x2 grid needs no distance tables:
// outputs (2 k + 1, 2 k + 2) share the taps (k, k + 1) at distances (0.25, -0.75) and (0.75, -0.25).
struct DeferredTile {
    uint32_t out_lo, out_hi;        // seg0: address of the tile's first owned output byte
    uint32_t dims;                  // nrow | ncol << 16: owned output rows / columns
    uint32_t misc;                  // rofs | cofs << 1 | edge << 2 | valid << 3 | full << 4 | lr0 << 8 | lc0 << 16
                                    //   rofs / cofs: the first owned row / column is the SECOND member of its pair (frame's top / left edge)
                                    //   edge: the hyper region reaches outside the frame (taps are clamped: edge hyper, zero image)
                                    //   full: 64 x 64 pairs, no single rows / columns -- no validity tests
                                    //   lr0 / lc0: first tap row / column of pair 0, hyper-region coordinates
    int rowpitch;                   // bytes per output row
    int hy0, hx0;                   // frame coordinates of the hyper region's origin (edge tiles)
    int i0, j0;                     // index of the first owned row / column in the frame's tables (float64 distances of the tie pass)
    int pad_[7];
};
static_assert(sizeof(DeferredTile) == 64, "two records in the control block");
__device__ __forceinline__ int dt_nrow(const uint32_t dims) { return (int)(dims & 0xFFFFu); }
__device__ __forceinline__ int dt_ncol(const uint32_t dims) { return (int)(dims >> 16); }

#if LERF_FUSED_NT == 512
#define LERF_PERSIST_ATTR __attribute__((amdgpu_waves_per_eu(1, 2)))       // 8 waves per CU: up to 256 VGPRs per thread
#else
#define LERF_PERSIST_ATTR
#endif
__global__ void __launch_bounds__(NT) LERF_PERSIST_ATTR
sr_persist_kernel(Params P, uint32_t* __restrict__ scratch_all, int total) {
    using D = Dims<2, false, false>;
    using PD = PersistDims;
    static_assert(CH == 3, "RGB instance");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    uint8_t* Bt = smem + D::OFF_B;
    int* ctl = reinterpret_cast<int*>(smem + PD::OFF_CTL);                    // [0..31] search results / counters, then two DeferredTile
    DeferredTile* dt = reinterpret_cast<DeferredTile*>(ctl + 32);
    uint32_t* tq = reinterpret_cast<uint32_t*>(smem + PD::OFF_TQ);
    int* tq_count = ctl + 23;
    uint32_t* __restrict__ scr = scratch_all + (size_t)blockIdx.x * PD::SCR_DWORDS;

    const int tiles = P.tiles_y * P.tiles_x;
    const int G = (int)gridDim.x;
    const int H = P.H, W = P.W;
    const float gscale = s3::gauss_scale(P.max_sigma);
    int cur = 0;                                                                // DeferredTile slot of the tile being looked up
    bool have_prev = false;
    if (tid == 0) { dt[0].misc = 0; dt[1].misc = 0; *tq_count = 0; }

    // ------------------------------------------------------------------ deferred stage 3: one per-channel sub-task
    // sub-task s of the previous tile: block task t = tid + (s / 3) * NT, channel s % 3 (s, kk, c wave-uniform).
    struct SubTask { uint32_t d[4]; int il0, jl0; };                        // the four tap dwords [a * 2 + b] and the block's first output row / column
    // the hot part of a tile's record as wave-uniform scalars (read from LDS once per tile: an LDS read inside a sub-task would
    // wait for the wave's 36 piece stores -- LDS operations of a wave complete in order)
    struct Hot { uint32_t out_lo, out_hi, dims, misc; int pitch; };
    auto hot_of = [&](const DeferredTile& T) {
        Hot h;
        h.out_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.out_lo);
        h.out_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.out_hi);
        h.dims = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.dims);
        h.misc = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.misc);
        h.pitch = __builtin_amdgcn_readfirstlane(T.rowpitch);
        return h;
    };
    // taps of the block whose first tap sits at (y, x) of the hyper region, channel c
    auto load_taps = [&](const DeferredTile& T, uint32_t misc, int y, int x, int c, uint32_t (&d)[4]) {
        if (misc & 4u) {
            // tile at the frame's edge: positions outside the frame were never looked up -- the clamped position's hyper-parameters,
            // image byte 0 (what fill_outside() of sr_fused_kernel leaves in its LDS array)
            const int hy0 = T.hy0, hx0 = T.hx0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int gy = hy0 + y + b, gx = hx0 + x + a;
                    const int cy = clampi(gy, 0, H - 1), cx = clampi(gx, 0, W - 1);
                    const uint32_t v = scr[(cy - hy0) * D::HP + (cx - hx0) * CH + c];
                    d[a * 2 + b] = (cy != gy || cx != gx) ? (v & 0x00FFFFFFu) : v;
                }
        } else {
            const uint32_t* p = scr + (y * D::HP + x * CH + c);
            d[0] = p[0]; d[1] = p[D::HP]; d[2] = p[CH]; d[3] = p[D::HP + CH];
        }
    };
    auto tie_eval = [&](const DeferredTile& T, int il, int xc) -> uint8_t {
        const uint32_t misc = T.misc;
        const int jl = xc / CH, c = xc - jl * CH;
        const int lr = (int)((misc >> 8) & 0xFFu) + ((il + (int)(misc & 1u)) >> 1), lc = (int)((misc >> 16) & 0xFFu) + ((jl + (int)((misc >> 1) & 1u)) >> 1);
        uint32_t dd[4];
        double dx64[2], dy64[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) dx64[b] = P.dis_r64[(int64_t)(T.i0 + il) * 2 + b];
#pragma unroll
        for (int a = 0; a < 2; ++a) dy64[a] = P.dis_c64[(int64_t)(T.j0 + jl) * 2 + a];
        load_taps(T, misc, lr, lc, c, dd);
        return (uint8_t)s3::to_u8_d(s3::eval64<true, 2>(dd, dx64, dy64, P.max_sigma));
    };
    // first half: where the block is, and its tap loads in flight (issued in FRONT of the wave's piece stores)
    auto sub_issue = [&](const DeferredTile& T, const Hot& Hh, int s, SubTask& X) -> bool {
        const int kk = s / 3, c = s - kk * 3;
        const uint32_t dims = Hh.dims, misc = Hh.misc;
        const int rofs = (int)(misc & 1u), cofs = (int)((misc >> 1) & 1u);
        const int ncg = (dt_ncol(dims) + cofs + 1) >> 1;
        const int nblk = ((dt_nrow(dims) + rofs + 1) >> 1) * ncg;
        const int t = tid + kk * NT;
        if (t >= nblk) return false;
        int g, h;
        if (ncg == 64) { g = t >> 6; h = t & 63; } else { g = t / ncg; h = t - g * ncg; }
        X.il0 = 2 * g - rofs;
        X.jl0 = 2 * h - cofs;
        load_taps(T, misc, (int)((misc >> 8) & 0xFFu) + g, (int)((misc >> 16) & 0xFFu) + h, c, X.d);
        return true;
    };
    // second half: the arithmetic of run_blocks (lerf_fused_impl.h) for one channel, packed over the two COLUMNS of the block,
    // and the four output bytes.  RECOVER = false: the normal pass (outputs stored; ties queued, a full queue only counts
    // them).  RECOVER = true: the pass behind a queue overflow (lerf_sr_geo_t.tie_queue_cap, a test hook: 2048 entries hold a
    // tile's ~15 ties) -- nothing but the float64 value of every tie is stored.  The float64 evaluation never sits inside the
    // phase loop.
    auto sub_finish = [&](const DeferredTile& T, const Hot& Hh, int s, const SubTask& X, auto recover_c) {
        constexpr bool RECOVER = decltype(recover_c)::value;
        const int c = s % 3;
        const uint32_t dims = Hh.dims, misc = Hh.misc;
        const float qa = 0.25f * gscale, qb = 0.75f * gscale;               // the table values 0.25 / 0.75 times the scale: geo_stage's products
        s3::f2 DX[2], DY[2];                                                // [tap] (first member of the pair, second member)
        DX[0].x = qa; DX[0].y = qb; DX[1].x = -qb; DX[1].y = -qa;
        DY[0] = DX[0]; DY[1] = DX[1];
        s3::f2 NUM[2], DEN[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t d = X.d[a * 2 + b];
                s3::f2 V, K1, K2, M2;
                V.x = V.y = (float)(d >> 24);
                K1.x = (float)((d >> 8) & 0xFFu);
                M2.x = s3::gauss_m2rho_u8((float)(d & 0xFFu));
                K2.x = (float)((d >> 16) & 0xFFu);
                const s3::f2 TYV = s3::pk_mul_blo(DY[a], K2);
                const s3::f2 P0 = s3::pk_mul_blo(TYV, M2);
                const s3::f2 TY2 = s3::pk_mul(TYV, TYV);
                const s3::f2 TX = s3::pk_mul_blo(DX[b], K1);
                const s3::f2 E0 = s3::pk_fma_ab<false>(TX, P0, s3::pk_fma_aa<false>(TX, TY2));
                const s3::f2 E1 = s3::pk_fma_ab<true>(TX, P0, s3::pk_fma_aa<true>(TX, TY2));
                s3::f2 W0, W1;
                W0.x = __builtin_amdgcn_exp2f(-E0.x); W0.y = __builtin_amdgcn_exp2f(-E0.y);
                W1.x = __builtin_amdgcn_exp2f(-E1.x); W1.y = __builtin_amdgcn_exp2f(-E1.y);
                if (a == 0 && b == 0) {
                    DEN[0] = W0; DEN[1] = W1;
                    NUM[0] = W0 * V; NUM[1] = W1 * V;
                } else {
                    NUM[0] = __builtin_elementwise_fma(W0, V, NUM[0]); NUM[1] = __builtin_elementwise_fma(W1, V, NUM[1]);
                    DEN[0] = DEN[0] + W0; DEN[1] = DEN[1] + W1;
                }
            }
        const int pitch = Hh.pitch;
        uint8_t* ob = reinterpret_cast<uint8_t*>(((uint64_t)Hh.out_hi << 32) | (uint64_t)Hh.out_lo) + ((int64_t)X.il0 * pitch + X.jl0 * CH + c);
        const bool full = (misc & 16u) != 0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float den = q ? DEN[r].y : DEN[r].x, num = q ? NUM[r].y : NUM[r].x;
                const float xf = s3::finish_div(num, den);
                const float rr = __builtin_rintf(xf);
                const bool have = full || ((unsigned)(X.il0 + r) < (unsigned)dt_nrow(dims) && (unsigned)(X.jl0 + q) < (unsigned)dt_ncol(dims));
                if (have) {
                    uint8_t* o = ob + (r * pitch + q * CH);
                    const bool tie = __builtin_fabsf(xf - rr) > 0.5f - s3::kTieEps && P.dis_r64 != nullptr;
                    if constexpr (RECOVER) {
                        if (tie) *o = tie_eval(T, X.il0 + r, (X.jl0 + q) * CH + c);
                    } else {
                        *o = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(rr, 0u, 0u);
                        if (tie) {
                            // rounding tie: queued for the float64 pass behind the tile's last sub-task
                            const int slot = atomicAdd(tq_count, 1);
                            if (slot < P.tq_cap) tq[slot] = ((uint32_t)(X.il0 + r) << 16) | (uint32_t)((X.jl0 + q) * CH + c);
                        }
                    }
                }
            }
    };
    auto sub_tasks_of = [&](const Hot& T) {
        const uint32_t dims = T.dims, misc = T.misc;
        const int ncg = (dt_ncol(dims) + (int)((misc >> 1) & 1u) + 1) >> 1;
        const int nblk = ((dt_nrow(dims) + (int)(misc & 1u) + 1) >> 1) * ncg;
        return (misc & 8u) ? ((nblk + NT - 1) / NT) * CH : 0;
    };
    // the float64 pass over the queued ties of a tile (all its sub-tasks done and their stores drained)
    auto tie_pass = [&](const DeferredTile& T) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nall = *tq_count;
        const int nq = min(nall, P.tq_cap);
        for (int i = tid; i < nq; i += NT) {
            const uint32_t e = tq[i];
            const int il = (int)(e >> 16), xc = (int)(e & 0xFFFFu);
            uint8_t* o = reinterpret_cast<uint8_t*>(((uint64_t)T.out_hi << 32) | (uint64_t)T.out_lo) + ((int64_t)il * T.rowpitch + xc);
            *o = tie_eval(T, il, xc);
        }
        if (nall > P.tq_cap) {
            // the queue overflowed: every thread walks its sub-tasks again and patches the ties itself
            const Hot Hh = hot_of(T);
            const int st = sub_tasks_of(Hh);
            for (int s = 0; s < st; ++s) {
                SubTask X;
                if (sub_issue(T, Hh, s, X)) sub_finish(T, Hh, s, X, std::true_type{});
            }
        }
        __syncthreads();
        if (tid == 0) *tq_count = 0;
    };

    const int tid0 = tid, wv0 = wv;
    for (int b = (int)blockIdx.x; b < total; b += G) {
        // Every per-thread constant of the tile body (list positions, DPP lane masks, copy addresses ...) is loop-invariant: the
        // compiler hoists them all in front of the tile loop and keeps them in registers across it -- 198 VGPR spills.  An opaque
        // copy of the thread index per iteration keeps the body what it is in sr_fused_kernel: recomputed per tile, dead at its end.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int wave = tid >> 6;
        const int wv = wv0;
        int bid = xcd_order(b, total);
        const int frame = bid / tiles;
        bid -= frame * tiles;
        const int tyi = bid / P.tiles_x, txi = bid - tyi * P.tiles_x;
        const int ty0 = P.ty_org + tyi * TH, tx0 = P.tx_org + txi * TW;
        const uint8_t* __restrict__ fsrc = P.feat + frame * P.feat_sn;
        uint8_t* __restrict__ outp = P.out + frame * P.out_sn;
        const int hy0 = ty0 - D::HR, hx0 = tx0 - D::HR;
        const int fy0 = ty0 - D::R3 - R2, fx0 = tx0 - D::R3 - R2;
        const bool interior = fy0 >= 0 && fx0 >= 0 && fy0 + D::FY <= H && fx0 + D::FX <= W;
        const int Hc = interior ? -1 : H, Wc = W;
        const DeferredTile& PT = dt[cur ^ 1];                                   // the previous tile (valid iff have_prev)
#ifdef LERF_STAMPS
        if (tid == 0) { P.stamps[(size_t)b * 16 + 8] = 0; P.stamps[(size_t)b * 16 + 9] = 0; P.stamps[(size_t)b * 16 + 15] = 0; }
#define LERF_PSTAMP(k) do { if (tid == 0) P.stamps[(size_t)b * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define LERF_PSTAMP_ADD(k, t0) do { if (tid == 0) P.stamps[(size_t)b * 16 + (k)] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
        if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); P.stamps[(size_t)b * 16 + 13] = t_; }
#else
#define LERF_PSTAMP(k) do {} while (0)
#define LERF_PSTAMP_ADD(k, t0) do {} while (0)
#endif
        LERF_PSTAMP(0);

        // ---- owned output rows / columns: four wave-parallel searches in the global tables, behind the feat-tile loads
        auto geo_search = [&]() {
            if (wave < 4) {
                const bool rows = wave < 2;
                const int* tab = rows ? P.left_r : P.left_c;
                const int n = rows ? P.oH : P.oW;
                const int ti = rows ? tyi : txi, tn = rows ? P.tiles_y : P.tiles_x, t0 = rows ? ty0 : tx0;
                int r;
                if (!(wave & 1)) r = ti == 0 ? 0 : wave_lower_bound(tab, n, t0 - D::R3, lane);
                else r = ti == tn - 1 ? n : wave_lower_bound(tab, n, t0 + (rows ? TH : TW) - D::R3, lane);
                if (lane == 0) ctl[16 + wave] = r;
            }
        };
        // ---- feat tile (with its halo) from the stage-1 launch
        {
            const bool dwords = interior && (D::FP & 3) == 0 && ((W * CH) & 3) == 0 && (reinterpret_cast<uintptr_t>(fsrc) & 3) == 0 && ((fx0 * CH) & 3) == 0;
            if (dwords) {
                constexpr int RD = D::FP / 4, ND = D::FY * RD, KD = (ND + NT - 1) / NT;
                const uint32_t* src = reinterpret_cast<const uint32_t*>(fsrc + ((int64_t)fy0 * W + fx0) * CH);
                const int rowdw = (W * CH) >> 2;
                uint32_t v[KD];
#pragma unroll
                for (int k = 0; k < KD; ++k) {
                    const int p = min(tid + k * NT, ND - 1);
                    const int ry = p / RD;
                    v[k] = __builtin_nontemporal_load(&src[(int64_t)ry * rowdw + (p - ry * RD)]);
                }
                geo_search();
#pragma unroll
                for (int k = 0; k < KD; ++k) {
                    const int p = tid + k * NT;
                    if (p < ND) reinterpret_cast<uint32_t*>(Bt)[p] = v[k];
                }
            } else {
                constexpr int KI = (D::NF + NT - 1) / NT;
                uint8_t v[KI];
#pragma unroll
                for (int k = 0; k < KI; ++k) {
                    const int p = min(tid + k * NT, D::NF - 1);
                    const int ry = p / D::FP;
                    const int r3 = p - ry * D::FP;
                    const int rx = r3 / CH;
                    const int gy = clampi(fy0 + ry, 0, H - 1), gx = clampi(fx0 + rx, 0, W - 1);
                    v[k] = fsrc[((int64_t)gy * W + gx) * CH + (r3 - rx * CH)];
                }
                geo_search();
#pragma unroll
                for (int k = 0; k < KI; ++k) {
                    const int p = tid + k * NT;
                    if (p < D::NF) Bt[p] = v[k];
                }
            }
        }
        __syncthreads();
        // ---- this tile's stage-3 record (one lane; an exact x2 grid -- lerf_sr_geo_t flag LERF_GEO_X2_TABLES vouches for it: pairs
        //      of rows / columns share their taps, g = i / 2 - 0.25 puts the first tap of a pair's first member 0.25 before it,
        //      of its second member 0.75)
        if (tid == 64) {
            DeferredTile& T = dt[cur];
            const int gi0 = ctl[16], gi1 = ctl[17], gj0 = ctl[18], gj1 = ctl[19];
            const bool any = gi1 > gi0 && gj1 > gj0;
            const float fr = any ? P.dis_r[(int64_t)gi0 * 2] : 0.0f, fc = any ? P.dis_c[(int64_t)gj0 * 2] : 0.0f;
            const int l_r = any ? P.left_r[gi0] : 0, l_c = any ? P.left_c[gj0] : 0;
            const int64_t rowpitch = P.out_pitch > 0 ? (int64_t)P.out_pitch : (int64_t)P.oW * CH;
            const uint64_t seg0 = (uint64_t)reinterpret_cast<uintptr_t>(outp + (int64_t)gi0 * rowpitch + (int64_t)gj0 * CH);
            T.out_lo = (uint32_t)seg0; T.out_hi = (uint32_t)(seg0 >> 32);
            T.rowpitch = (int)rowpitch;
            const int nrow = gi1 - gi0, ncol = gj1 - gj0;
            T.dims = (uint32_t)nrow | ((uint32_t)ncol << 16);
            T.i0 = gi0; T.j0 = gj0;
            T.hy0 = hy0; T.hx0 = hx0;
            const uint32_t rofs = fr > 0.5f ? 1u : 0u, cofs = fc > 0.5f ? 1u : 0u;
            const uint32_t edge = (hy0 < 0 || hx0 < 0 || hy0 + D::HY > H || hx0 + D::HX > W) ? 1u : 0u;
            const uint32_t full = (nrow == 2 * TH && ncol == 2 * TW && rofs == 0u && cofs == 0u) ? 1u : 0u;
            T.misc = rofs | (cofs << 1) | (edge << 2) | ((any ? 1u : 0u) << 3) | (full << 4) | ((uint32_t)(l_r - hy0) << 8) | ((uint32_t)(l_c - hx0) << 16);
        }
        LERF_PSTAMP(6);

        // ================================================================== stage 2 (as sr_fused_kernel, LeRF-G)
        constexpr int MAXR = D::MAXR;
        uint16_t* lst = reinterpret_cast<uint16_t*>(smem + D::OFF_LST);
        int* tab = reinterpret_cast<int*>(smem + D::OFF_TAB);
        for (int i = tid; i < MAXR * NT / 2; i += NT) reinterpret_cast<uint32_t*>(lst)[i] = 0xFFFFFFFFu;
        constexpr int KH = (D::NH + NT - 1) / NT;
        constexpr int KQ = (((D::NH + NT - 1) / NT) + 7) / 8;                   // registers of bin nibbles (8 positions each)
        uint32_t qn[KQ];
#pragma unroll
        for (int i = 0; i < KQ; ++i) qn[i] = 0;
        uint32_t xa, xb;
        {
            uint32_t hist = 0;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int p = tid * KH + k;
                uint32_t q = 15;
                if (p < D::NH) {
                    bool in = true;
                    const int a = center_addr<D::HP, D::FP>(p, hy0, hx0, fy0, fx0, Hc, Wc, &in);
                    if (in) q = bin_of_level((uint32_t)Bt[a] >> 4);
                }
                qn[k >> 3] |= q << (4 * (k & 7));
                hist += q < NBIN ? 1u << (8 * q) : 0u;
            }
            xa = (hist & 0xFFu) | ((hist << 8) & 0xFF0000u);
            xb = ((hist >> 16) & 0xFFu) | ((hist >> 8) & 0xFF0000u);
        }
        auto wave_scan = [](uint32_t v) {
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
            return v;
        };
        const uint32_t ia = wave_scan(xa), ib = wave_scan(xb);
        if (lane == 63) {
            int* t = tab + wave * 4;
            t[0] = (int)(ia & 0xFFFFu); t[1] = (int)(ia >> 16); t[2] = (int)(ib & 0xFFFFu); t[3] = (int)(ib >> 16);
        }
        __syncthreads();
        uint32_t ne_bins, cs_pack, ce_pack;
        uint32_t cur01, cur23;
        {
            const int bb = lane & 3;
            const int c = lane < NW * 4 ? tab[lane] : 0;                 // lane = w * 4 + bb, w < NW
            int incl = c;
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const int tot = __shfl(incl, (NW - 1) * 4 + bb);
            const int padded = (tot + 63) & ~63;
            int start = padded;
#pragma unroll
            for (int d = 1; d < 4; d <<= 1) {
                const int up = __shfl_up(start, d, 4);
                if (bb >= d) start += up;
            }
            start -= padded;
            const int base = start + incl - c;
            const int wl = wv * 4;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(base, wl), b1 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 1),
                           b2 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 2), b3 = (uint32_t)__builtin_amdgcn_readlane(base, wl + 3);
            const uint32_t ea = ia - xa, eb = ib - xb;
            cur01 = ((b0 + (ea & 0xFFFFu)) & 0xFFFFu) | ((b1 + (ea >> 16)) << 16);
            cur23 = ((b2 + (eb & 0xFFFFu)) & 0xFFFFu) | ((b3 + (eb >> 16)) << 16);
            const int sc = start >> 6, ec = (start + padded) >> 6;
            cs_pack = (uint32_t)__builtin_amdgcn_readlane(sc, 0) | ((uint32_t)__builtin_amdgcn_readlane(sc, 1) << 8) |
                      ((uint32_t)__builtin_amdgcn_readlane(sc, 2) << 16) | ((uint32_t)__builtin_amdgcn_readlane(sc, 3) << 24);
            ce_pack = (uint32_t)__builtin_amdgcn_readlane(ec, 0) | ((uint32_t)__builtin_amdgcn_readlane(ec, 1) << 8) |
                      ((uint32_t)__builtin_amdgcn_readlane(ec, 2) << 16) | ((uint32_t)__builtin_amdgcn_readlane(ec, 3) << 24);
            ne_bins = (uint32_t)(__ballot(tot > 0) & 0xFull);
        }
        // a thread's share of a piece: 16 bytes of every NW-th 1-KiB block (block NW i + wave).  1024 threads: 9 slabs; 512: 17
        constexpr int NSL = (PIECE_BLOCKS + NW - 1) / NW;
        uint4 pr[NSL];
#pragma unroll
        for (int i = 0; i < NSL; ++i) pr[i] = make_uint4(0, 0, 0, 0);
        const uint8_t* s2p = P.pack + (size_t)3 * LUT_PAD;
        constexpr int NL2 = 6;
        auto pre_load = [&](int l, int bq) {
            const uint4* s_ = reinterpret_cast<const uint4*>(s2p + ((size_t)l * NBIN + bq) * PIECE_BYTES) + (wave * 64 + lane);
#pragma unroll
            for (int i = 0; i < NSL; ++i) pr[i] = s_[i * NT];            // (the pack pads every piece to PIECE_BYTES: the last slab's loads stay inside it)
        };
        const int nbins = __builtin_popcount(ne_bins);
        const int nph = nbins * NL2;
        if (nph > 0) pre_load(0, __builtin_ctz(ne_bins));
        {
            const int p0 = tid * KH, ry0 = p0 / D::HP;
            uint32_t col = (uint32_t)(p0 - ry0 * D::HP);
            uint32_t ap = (uint32_t)((ry0 + D::HO) * D::FP + D::HO * CH) + col;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const uint32_t q = (qn[k >> 3] >> (4 * (k & 7))) & 0xFu;
                if (q < NBIN) {
                    const uint32_t pair = q < 2 ? cur01 : cur23;
                    const uint32_t v = (q & 1u) ? pair >> 16 : pair & 0xFFFFu;
                    lst[v] = (uint16_t)ap;
                    const uint32_t inc = (q & 1u) ? 0x10000u : 1u;
                    if (q < 2) cur01 += inc; else cur23 += inc;
                }
                ++ap;
                if (++col == (uint32_t)D::HP) { col = 0; ap += (uint32_t)(D::FP - D::HP); }
            }
        }
        __syncthreads();
        constexpr int MAXP = (MAXR + 1) / 2;
        uint32_t slot2[MAXP];
        uint32_t accA[MAXR], accB[MAXR];
        uint32_t wrounds = 0;
        uint32_t vmask = 0;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const uint32_t p = lst[k * NT + tid];
            uint32_t a = p;
            if (p != 0xFFFFu) vmask |= 1u << k;
            const unsigned long long real = __ballot(p != 0xFFFFu);
            if (real != 0ull) {
                const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)a, (int)__builtin_ctzll(real));
                if (p == 0xFFFFu) a = a1;
                wrounds |= 1u << k;
            }
            if (k & 1) slot2[k >> 1] |= a << 16; else slot2[k >> 1] = a;
            accA[k] = 0;
            accB[k] = 0;
        }
        __syncthreads();
        LERF_PSTAMP(7);

        // the previous tile's sub-tasks are dealt out over this tile's phases (an error-diffusion count: no division per phase)
        Hot PH = hot_of(PT);
        if (!have_prev) PH.misc = 0;
#ifdef LERF_PERSIST_NOS3
        PH.misc = 0;                                   // timing experiment only (wrong results): the persistent loop without any stage 3
#endif
        const int stotal = sub_tasks_of(PH);
        int sdone = 0, sacc = 0;
        // half of the waves store their part of the piece first and compute then, the other half the other way round: one half's
        // VALU work runs under the other half's LDS writes (a wave issues in order: its own stores and its own arithmetic do not overlap)
        const bool math_first = wv >= NW / 2;
        auto pre_store_issue = [&]() {
            // slab i = block NW i + wave at OFF_X + (NW i + wave) * 1024; the DS offset field is 16 bits, M0 moves in 64-KiB windows
            const uint32_t m0a = __builtin_amdgcn_readfirstlane((uint32_t)D::OFF_X + (uint32_t)wave * 1024u);
            constexpr int PER_WIN = 65536 / (NW * 1024);                 // slabs per window (4 at 1024 threads, 8 at 512)
#define LERF_PSLAB(I, O)                                                                                   \
            if constexpr ((I) < NSL) {                                                                         \
                /* M0 must stay below 128 KiB: the third window sits at +73728 and uses the top of the offset range */ \
                if ((I) % PER_WIN == 0) LERF_SET_M0(m0a + ((I) / PER_WIN < 2 ? (uint32_t)((I) / PER_WIN) * 65536u : 131072u - 57344u)); \
                if (NW * (I) + NW - 1 < PIECE_BLOCKS || NW * (I) + wave < PIECE_BLOCKS) {                      \
                    const uint4& r_ = pr[(I) < NSL ? (I) : 0];                                                 \
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(r_.x), "n"((O)) : "memory");          \
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(r_.y), "n"((O) + 256) : "memory");    \
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(r_.z), "n"((O) + 512) : "memory");    \
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(r_.w), "n"((O) + 768) : "memory");    \
                }                                                                                              \
            }
            if constexpr (NW == 16) {
                LERF_PSLAB(0, 0) LERF_PSLAB(1, 16384) LERF_PSLAB(2, 32768) LERF_PSLAB(3, 49152)
                LERF_PSLAB(4, 0) LERF_PSLAB(5, 16384) LERF_PSLAB(6, 32768) LERF_PSLAB(7, 49152)
                LERF_PSLAB(8, 57344)
            } else {
                static_assert(NW == 8 || NW == 16, "512 or 1024 threads");
                LERF_PSLAB(0, 0) LERF_PSLAB(1, 8192) LERF_PSLAB(2, 16384) LERF_PSLAB(3, 24576) LERF_PSLAB(4, 32768) LERF_PSLAB(5, 40960) LERF_PSLAB(6, 49152) LERF_PSLAB(7, 57344)
                LERF_PSLAB(8, 0) LERF_PSLAB(9, 8192) LERF_PSLAB(10, 16384) LERF_PSLAB(11, 24576) LERF_PSLAB(12, 32768) LERF_PSLAB(13, 40960) LERF_PSLAB(14, 49152) LERF_PSLAB(15, 57344)
                LERF_PSLAB(16, 57344)
            }
#undef LERF_PSLAB
        };
        int bq = 0, bq_next = 0, l = 0, bi = 0;
        uint32_t ne_left = ne_bins;
        uint32_t act = 0, qbase = 0;
        for (int ph = 0; ph < nph; ++ph) {
            if (l == 0) {
                bq = __builtin_ctz(ne_left);
                ne_left &= ne_left - 1u;
                bq_next = ne_left != 0u ? __builtin_ctz(ne_left) : 0;
                const int cs = (int)((cs_pack >> (8 * bq)) & 0xFFu), ce = (int)((ce_pack >> (8 * bq)) & 0xFFu);
                // list entry i sits in round i / NT of thread i % NT: chunk c (64 entries) = round c / NW of wave c % NW
                const int klo = cs > wv ? (cs - wv + NW - 1) / NW : 0, khi = ce > wv ? (ce - wv + NW - 1) / NW : 0;
                act = wrounds & ((1u << khi) - 1u) & ~((1u << klo) - 1u);
                qbase = lds_addr(smem + D::OFF_X) - (uint32_t)bin_lo(bq) * (kStrideA * 4u);
            }
            const unsigned long long t_copy = LERF_NOW();
            (void)t_copy;
            // ---- deferred stage 3 of the previous tile: this phase's share of its sub-tasks.  The tap loads go out first, the wave's
            //      piece stores behind them; the arithmetic runs while the piece stores (this wave's and the other waves') occupy the LDS
            sacc += stotal;
            int n_here = 0;
            while (sacc >= nph) { sacc -= nph; ++n_here; }
            if (ph == nph - 1) n_here = stotal - sdone;
#if LERF_PERSIST_ORDER == 1
            // v2: half of the waves compute first and store then (the prefetch registers stay live under the arithmetic: spills)
            SubTask X;
            bool on = false;
            if (n_here > 0) on = sub_issue(PT, PH, sdone, X);
            if (!math_first) pre_store_issue();
            if (n_here > 0) {
                if (on) sub_finish(PT, PH, sdone, X, std::false_type{});
                for (int s = 1; s < n_here; ++s) {
                    SubTask Y;
                    if (sub_issue(PT, PH, sdone + s, Y)) sub_finish(PT, PH, sdone + s, Y, std::false_type{});
                }
                sdone += n_here;
            }
            if (math_first) pre_store_issue();
#else
            // v3: every wave stores its part of the piece first -- the 36 prefetch registers are dead before the sub-task needs any
            (void)math_first;
            pre_store_issue();
            for (int s = 0; s < n_here; ++s) {
                SubTask Y;
                if (sub_issue(PT, PH, sdone + s, Y)) sub_finish(PT, PH, sdone + s, Y, std::false_type{});
            }
            sdone += n_here;
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            Off3 o0, o1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                o0.o[i] = P.s2off[l][i];
                o1.o[i] = P.s2off[l][3 + i];
            }
            const uint32_t bt_a = lds_addr(Bt);
            const unsigned st_a = kStrideA * 4, st_b = kStrideB * 4, st_c = kStrideC * 4, st_d = kStrideD * 4;
            __syncthreads();
            if (l < NL2 - 1) pre_load(l + 1, bq);
            else if (bi + 1 < nbins) pre_load(0, bq_next);
            LERF_PSTAMP_ADD(8, t_copy);
            const unsigned long long t_look = LERF_NOW();
            (void)t_look;
#pragma unroll
            for (int k = 0; k < MAXR; ++k) {
                if ((act >> k) & 1u) {
                    const uint32_t sa = (k & 1) ? (slot2[k >> 1] >> 16) : (slot2[k >> 1] & 0xFFFFu);
                    const uint32_t cpa = bt_a + sa;
                    uint32_t ra = lds_pixel_hi(cpa);
                    uint32_t rb0 = lds_pixel_hi(cpa + (uint32_t)o0.o[0]), rc0 = lds_pixel_hi(cpa + (uint32_t)o0.o[1]),
                             rd0 = lds_pixel_hi(cpa + (uint32_t)o0.o[2]);
                    uint32_t rb1 = lds_pixel_hi(cpa + (uint32_t)o1.o[0]), rc1 = lds_pixel_hi(cpa + (uint32_t)o1.o[1]),
                             rd1 = lds_pixel_hi(cpa + (uint32_t)o1.o[2]);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb0), "+v"(rc0), "+v"(rd0), "+v"(rb1), "+v"(rc1), "+v"(rd1));
                    const int basea = (int)(__umul24(msb_of(ra), kStrideA * 4) + qbase);
                    const unsigned ka = key_of(ra, st_a);
                    const Walk<4> W0 = simplex_walk<4>(ka, basea, rb0, rc0, rd0, st_b, st_c, st_d);
                    const Walk<4> W1 = simplex_walk<4>(ka, basea, rb1, rc1, rd1, st_b, st_c, st_d);
                    uint32_t d0[5], d1[5];
#pragma unroll
                    for (int n = 0; n < 5; ++n) d0[n] = W0.ld32(n);
#pragma unroll
                    for (int n = 0; n < 5; ++n) d1[n] = W1.ld32(n);
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
                    const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
                    uint32_t a = accA[k], bb = accB[k];
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        a += __umul24(w0[n], d0[n]);
                        bb = mad_hi16(w0[n], d0[n], bb);
                    }
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        a += __umul24(w1[n], d1[n]);
                        bb = mad_hi16(w1[n], d1[n], bb);
                    }
                    accA[k] = a;
                    accB[k] = bb;
                }
            }
            __syncthreads();
            LERF_PSTAMP_ADD(9, t_look);
            if (++l == NL2) { l = 0; ++bi; }
        }
        LERF_PSTAMP(10);
        // ---- finalise: hq = rne(clip(N/192 + 127)) -> the tile's dwords, first into LDS (over the dead piece): the accumulators
        //      are dead before the float64 code of the previous tile's tie pass needs the registers
        uint32_t* Dl = reinterpret_cast<uint32_t*>(smem + D::OFF_X);
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            if ((vmask >> k) & 1u) {
                constexpr int div2 = kQ * 12;
                const int n0 = (int)(accA[k] & 0xFFFFu) - div2;
                const int n2 = (int)(accA[k] >> 16) - div2;
                const int n1 = (int)((accB[k] - (accA[k] >> 16)) >> 8) - div2;
                const uint32_t h0 = (uint32_t)rne_div_clip255_fast(n0, div2);
                const uint32_t h1 = (uint32_t)rne_div_clip255_fast(n1, div2);
                const uint32_t h2 = (uint32_t)rne_div_clip255_fast(n2, div2);
                const uint32_t a = (k & 1) ? (slot2[k >> 1] >> 16) : (slot2[k >> 1] & 0xFFFFu);
                const uint32_t row = a / (uint32_t)D::FP;
                const uint32_t p = a - row * (uint32_t)(D::FP - D::HP) - (uint32_t)(D::HO * D::HP + D::HO * CH);
                Dl[p] = h0 | (h1 << 8) | (h2 << 16) | ((uint32_t)Bt[a] << 24);
            }
        }
        LERF_PSTAMP(11);
        // ---- whatever is left of the previous tile (fewer phases than sub-tasks), then its tie pass: the scratch is about to be
        //      overwritten
        if (stotal > 0) {
            for (; sdone < stotal; ++sdone) {
                SubTask X;
                if (sub_issue(PT, PH, sdone, X)) sub_finish(PT, PH, sdone, X, std::false_type{});
            }
            tie_pass(PT);
        } else {
            __syncthreads();
        }
        // ---- LDS -> the workgroup's scratch (positions outside the frame carry nothing: the sub-tasks clamp their taps)
        for (int p = tid; p < D::NH; p += NT) scr[p] = Dl[p];
        // the scratch written by all waves is read by all waves of the next iteration: one workgroup = one CU = one L1 -- a
        // workgroup-scope release / acquire (an agent-scope release writes the XCD's whole L2 back: 100 k cycles per tile, measured)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        LERF_PSTAMP(12);
#ifdef LERF_STAMPS
        if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); P.stamps[(size_t)b * 16 + 14] = t_; }
#endif
        have_prev = true;
        cur ^= 1;
    }
    // ---- the last tile of this workgroup: its stage 3 without anything to hide under
    if (have_prev) {
        const DeferredTile& PT = dt[cur ^ 1];
        const Hot PH = hot_of(PT);
#ifdef LERF_PERSIST_NOS3
        const int stotal = 0;
#else
        const int stotal = sub_tasks_of(PH);
#endif
        if (stotal > 0) {
            for (int s = 0; s < stotal; ++s) {
                SubTask X;
                if (sub_issue(PT, PH, s, X)) sub_finish(PT, PH, s, X, std::false_type{});
            }
            tie_pass(PT);
        }
    }
#undef LERF_PSTAMP
#undef LERF_PSTAMP_ADD
}
