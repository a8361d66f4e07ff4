// Micro-benchmark: the stage-2 lookup slot of sr_fused_kernel (two simplex walks + ten dword gathers + MACs) replayed
// in isolation on random LDS contents, to separate what the VALU mix costs from what the phase structure costs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I lerf-pytorch_amd/csrc tools/ubench/s2_slot.hip -o /tmp/s2_slot
// Variants: V=0 the slot as in the kernel (round test + padding test); V=1 no tests; V=2 no LDS at all (loads replaced
// by register moves); V=3 two slots interleaved by hand (loads of both, walks of both, gathers of both, MACs of both).
#include "../../lerf-pytorch_amd/csrc/lerf_fused.hip"
#include <cstdio>
#include <vector>

using namespace lerf;
using namespace lerf::fused;

constexpr int ROUNDS = 16;

template <int V>
__device__ __forceinline__ void slot(uint32_t cpa, const Off3& o0, const Off3& o1, uint32_t qbase, unsigned st_a, unsigned st_b,
                                     unsigned st_c, unsigned st_d, uint32_t& accA, uint32_t& accB) {
    uint32_t ra, rb0, rc0, rd0, rb1, rc1, rd1;
    if (V == 2) {
        ra = cpa * 0x9E3779B1u; rb0 = ra + o0.o[0]; rc0 = ra ^ o0.o[1]; rd0 = ra + o0.o[2]; rb1 = ra ^ o1.o[0]; rc1 = ra + o1.o[1]; rd1 = ra ^ o1.o[2];
    } else {
        ra = lds_pixel_hi(cpa);
        rb0 = lds_pixel_hi(cpa + (uint32_t)o0.o[0]); rc0 = lds_pixel_hi(cpa + (uint32_t)o0.o[1]); rd0 = lds_pixel_hi(cpa + (uint32_t)o0.o[2]);
        rb1 = lds_pixel_hi(cpa + (uint32_t)o1.o[0]); rc1 = lds_pixel_hi(cpa + (uint32_t)o1.o[1]); rd1 = lds_pixel_hi(cpa + (uint32_t)o1.o[2]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb0), "+v"(rc0), "+v"(rd0), "+v"(rb1), "+v"(rc1), "+v"(rd1));
    }
    const int basea = (int)(__umul24(msb_of(ra) & 3u, kStrideA * 4) + qbase);
    const unsigned ka = key_of(ra, st_a);
    const Walk<4> W0 = simplex_walk<4>(ka, basea, rb0, rc0, rd0, st_b, st_c, st_d);
    const Walk<4> W1 = simplex_walk<4>(ka, basea, rb1, rc1, rd1, st_b, st_c, st_d);
    uint32_t d0[5], d1[5];
    if (V == 8 || V == 9) {
        asm volatile("ds_read_b32 %0, %1" : "=v"(d0[0]) : "v"(W0.a(0)));
        asm volatile("ds_read_b32 %0, %1" : "=v"(d0[1]) : "v"(W0.a(1)));
        asm volatile("ds_read_b32 %0, %1" : "=v"(d0[2]) : "v"(W0.a(2)));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d0[3]) : "v"(W0.a(3)), "n"(Walk<4>::ALL));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d0[4]) : "v"(W0.a(4)), "n"(Walk<4>::ALL));
        asm volatile("ds_read_b32 %0, %1" : "=v"(d1[0]) : "v"(W1.a(0)));
        asm volatile("ds_read_b32 %0, %1" : "=v"(d1[1]) : "v"(W1.a(1)));
        asm volatile("ds_read_b32 %0, %1" : "=v"(d1[2]) : "v"(W1.a(2)));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d1[3]) : "v"(W1.a(3)), "n"(Walk<4>::ALL));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d1[4]) : "v"(W1.a(4)), "n"(Walk<4>::ALL));
        const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
        const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
        uint32_t a = accA, bb = accB;
        if (V == 8) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(d0[0]), "+v"(d0[1]), "+v"(d0[2]), "+v"(d0[3]), "+v"(d0[4]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0[0]), "+v"(d0[1]), "+v"(d0[2]), "+v"(d0[3]), "+v"(d0[4]), "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d1[3]), "+v"(d1[4]));
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w0[n], d0[n]); bb += __umul24(w0[n], d0[n] >> 24); }
        if (V == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d1[3]), "+v"(d1[4]));
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w1[n], d1[n]); bb += __umul24(w1[n], d1[n] >> 24); }
        accA = a; accB = bb;
        return;
    }
#pragma unroll
    for (int n = 0; n < 5; ++n) d0[n] = V == 2 ? W0.a(n) : (V == 5 || V == 7) ? lds_ld32(qbase + (threadIdx.x & 63) * 4 + n * 256 + (W0.a(n) >> 30)) : W0.ld32(n);
#pragma unroll
    for (int n = 0; n < 5; ++n) d1[n] = V == 2 ? W1.a(n) : (V == 5 || V == 7) ? lds_ld32(qbase + (threadIdx.x & 63) * 4 + n * 256 + 2048 + (W1.a(n) >> 30)) : W1.ld32(n);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
    const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
    uint32_t a = accA, bb = accB;
#pragma unroll
    for (int n = 0; n < 5; ++n) { a += __umul24(w0[n], d0[n]); bb += __umul24(w0[n], d0[n] >> 24); }
#pragma unroll
    for (int n = 0; n < 5; ++n) { a += __umul24(w1[n], d1[n]); bb += __umul24(w1[n], d1[n] >> 24); }
    accA = a; accB = bb;
}

template <int V, int NTH>
__global__ void __launch_bounds__(1024) k(uint32_t* out, unsigned long long* cyc, int iters, int rsq, int req, Off3 o0, Off3 o1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    // random LDS contents: feat tile at 0 (16 KB), piece at 16 KB (96 KB)
    for (int i = tid; i < (16384 + PIECE_LDS) / 4; i += NTH) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t bt_a0 = lds_addr(smem), qbase = lds_addr(smem + 16384);
    const unsigned st_a = kStrideA * 4, st_b = kStrideB * 4, st_c = kStrideC * 4, st_d = kStrideD * 4;
    uint32_t sa[ROUNDS], accA[ROUNDS], accB[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        sa[r] = 4 * 216 + (((V >= 6 ? tid : tid * 7) + r * 1031 + blockIdx.x * 13) % (64 * 216));    // inside the tile, away from its edge
        accA[r] = 0; accB[r] = 0;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t bt_a = bt_a0 + (uint32_t)(it & 1) * 3u;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (V == 0) {
                if (r >= rsq && r < req) {
                    if (sa[r] != 0xFFFFu) slot<0>(bt_a + sa[r], o0, o1, qbase, st_a, st_b, st_c, st_d, accA[r], accB[r]);
                }
            } else {
                slot<V>(bt_a + sa[r], o0, o1, qbase, st_a, st_b, st_c, st_d, accA[r], accB[r]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) x ^= accA[r] + accB[r];
    out[blockIdx.x * NTH + tid] = x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


// ---- V=3: software-pipelined rounds: the pixel loads of round r+1 are issued behind the gathers of round r, so that
//      their LDS round trip overlaps the MACs.  Every LDS read is inline asm with hand-placed waits.
__device__ __forceinline__ uint32_t gather32(uint32_t a) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a)); return r; }
template <int OFF>
__device__ __forceinline__ uint32_t gather32o(uint32_t a) { uint32_t r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
struct Px { uint32_t a, b0, c0, d0, b1, c1, d1; };
__device__ __forceinline__ void px_issue(Px& p, uint32_t cpa, const Off3& o0, const Off3& o1) {
    p.a = lds_pixel_hi(cpa);
    p.b0 = lds_pixel_hi(cpa + (uint32_t)o0.o[0]); p.c0 = lds_pixel_hi(cpa + (uint32_t)o0.o[1]); p.d0 = lds_pixel_hi(cpa + (uint32_t)o0.o[2]);
    p.b1 = lds_pixel_hi(cpa + (uint32_t)o1.o[0]); p.c1 = lds_pixel_hi(cpa + (uint32_t)o1.o[1]); p.d1 = lds_pixel_hi(cpa + (uint32_t)o1.o[2]);
}
#define WAITN(N, R) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R))
template <int NTH>
__global__ void __launch_bounds__(1024) kp(uint32_t* out, unsigned long long* cyc, int iters, int rsq, int req, Off3 o0, Off3 o1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < (16384 + PIECE_LDS) / 4; i += NTH) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t bt_a0 = lds_addr(smem), qbase = lds_addr(smem + 16384);
    const unsigned st_a = kStrideA * 4, st_b = kStrideB * 4, st_c = kStrideC * 4, st_d = kStrideD * 4;
    uint32_t sa[ROUNDS], accA[ROUNDS], accB[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        sa[r] = 4 * 216 + ((tid * 7 + r * 1031 + blockIdx.x * 13) % (64 * 216));
        accA[r] = 0; accB[r] = 0;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t bt_a = bt_a0 + (uint32_t)(it & 1) * 3u;
        Px P;
        px_issue(P, bt_a + sa[0], o0, o1);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= rsq && r < req) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.a), "+v"(P.b0), "+v"(P.c0), "+v"(P.d0), "+v"(P.b1), "+v"(P.c1), "+v"(P.d1));
                const int basea = (int)(__umul24(msb_of(P.a) & 3u, kStrideA * 4) + qbase);
                const unsigned ka = key_of(P.a, st_a);
                const Walk<4> W0 = simplex_walk<4>(ka, basea, P.b0, P.c0, P.d0, st_b, st_c, st_d);
                const Walk<4> W1 = simplex_walk<4>(ka, basea, P.b1, P.c1, P.d1, st_b, st_c, st_d);
                if (r + 1 < ROUNDS) { if (r + 1 < req) px_issue(P, bt_a + sa[r + 1 < ROUNDS ? r + 1 : r], o0, o1); }
                uint32_t d0[5], d1[5];
#pragma unroll
                for (int n = 0; n < 5; ++n) d0[n] = W0.ld32(n);
#pragma unroll
                for (int n = 0; n < 5; ++n) d1[n] = W1.ld32(n);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
                const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
                uint32_t a = accA[r], bb = accB[r];
#pragma unroll
                for (int n = 0; n < 5; ++n) { a += __umul24(w0[n], d0[n]); bb += __umul24(w0[n], d0[n] >> 24); }
#pragma unroll
                for (int n = 0; n < 5; ++n) { a += __umul24(w1[n], d1[n]); bb += __umul24(w1[n], d1[n] >> 24); }
                accA[r] = a; accB[r] = bb;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) x ^= accA[r] + accB[r];
    out[blockIdx.x * NTH + tid] = x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- V=4: deep pipeline: round r's gathers are in flight while round r+1's pixels are fetched and walked; the MACs of
//      round r come last.  LDS order per iteration: px(r+1), gathers(r); wait for px only; walk(r+1); MACs(r).
template <int NTH>
__global__ void __launch_bounds__(1024) kd(uint32_t* out, unsigned long long* cyc, int iters, int rsq, int req, Off3 o0, Off3 o1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < (16384 + PIECE_LDS) / 4; i += NTH) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t bt_a0 = lds_addr(smem), qbase = lds_addr(smem + 16384);
    const unsigned st_a = kStrideA * 4, st_b = kStrideB * 4, st_c = kStrideC * 4, st_d = kStrideD * 4;
    uint32_t sa[ROUNDS], accA[ROUNDS], accB[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        sa[r] = 4 * 216 + ((tid * 7 + r * 1031 + blockIdx.x * 13) % (64 * 216));
        accA[r] = 0; accB[r] = 0;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t bt_a = bt_a0 + (uint32_t)(it & 1) * 3u;
        Px P;
        px_issue(P, bt_a + sa[0], o0, o1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.a), "+v"(P.b0), "+v"(P.c0), "+v"(P.d0), "+v"(P.b1), "+v"(P.c1), "+v"(P.d1));
        Walk<4> W0, W1;
        {
            const int basea = (int)(__umul24(msb_of(P.a) & 3u, kStrideA * 4) + qbase);
            const unsigned ka = key_of(P.a, st_a);
            W0 = simplex_walk<4>(ka, basea, P.b0, P.c0, P.d0, st_b, st_c, st_d);
            W1 = simplex_walk<4>(ka, basea, P.b1, P.c1, P.d1, st_b, st_c, st_d);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r >= rsq && r < req) {
                const bool more = r + 1 < ROUNDS && r + 1 < req;
                if (more) px_issue(P, bt_a + sa[r + 1 < ROUNDS ? r + 1 : r], o0, o1);
                uint32_t d0[5], d1[5];
                d0[0] = gather32(W0.a(0)); d0[1] = gather32(W0.a(1)); d0[2] = gather32(W0.a(2));
                d0[3] = gather32o<Walk<4>::ALL>(W0.a(3)); d0[4] = gather32o<Walk<4>::ALL>(W0.a(4));
                d1[0] = gather32(W1.a(0)); d1[1] = gather32(W1.a(1)); d1[2] = gather32(W1.a(2));
                d1[3] = gather32o<Walk<4>::ALL>(W1.a(3)); d1[4] = gather32o<Walk<4>::ALL>(W1.a(4));
                const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
                const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
                if (more) {
                    asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(P.a), "+v"(P.b0), "+v"(P.c0), "+v"(P.d0), "+v"(P.b1), "+v"(P.c1), "+v"(P.d1));
                    const int basea = (int)(__umul24(msb_of(P.a) & 3u, kStrideA * 4) + qbase);
                    const unsigned ka = key_of(P.a, st_a);
                    W0 = simplex_walk<4>(ka, basea, P.b0, P.c0, P.d0, st_b, st_c, st_d);
                    W1 = simplex_walk<4>(ka, basea, P.b1, P.c1, P.d1, st_b, st_c, st_d);
                }
                uint32_t a = accA[r], bb = accB[r];
                WAITN(9, d0[0]); a += __umul24(w0[0], d0[0]); bb += __umul24(w0[0], d0[0] >> 24);
                WAITN(8, d0[1]); a += __umul24(w0[1], d0[1]); bb += __umul24(w0[1], d0[1] >> 24);
                WAITN(7, d0[2]); a += __umul24(w0[2], d0[2]); bb += __umul24(w0[2], d0[2] >> 24);
                WAITN(6, d0[3]); a += __umul24(w0[3], d0[3]); bb += __umul24(w0[3], d0[3] >> 24);
                WAITN(5, d0[4]); a += __umul24(w0[4], d0[4]); bb += __umul24(w0[4], d0[4] >> 24);
                WAITN(4, d1[0]); a += __umul24(w1[0], d1[0]); bb += __umul24(w1[0], d1[0] >> 24);
                WAITN(3, d1[1]); a += __umul24(w1[1], d1[1]); bb += __umul24(w1[1], d1[1] >> 24);
                WAITN(2, d1[2]); a += __umul24(w1[2], d1[2]); bb += __umul24(w1[2], d1[2] >> 24);
                WAITN(1, d1[3]); a += __umul24(w1[3], d1[3]); bb += __umul24(w1[3], d1[3] >> 24);
                WAITN(0, d1[4]); a += __umul24(w1[4], d1[4]); bb += __umul24(w1[4], d1[4] >> 24);
                accA[r] = a; accB[r] = bb;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) x ^= accA[r] + accB[r];
    out[blockIdx.x * NTH + tid] = x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int NTH>
void run(const char* name, uint32_t* d, unsigned long long* c) {
    const int iters = 200, blocks = 256;
    const int lds = 16384 + PIECE_LDS + 4096;       // one workgroup per CU
    Off3 o0 = tile_offsets<216>('s', 0), o1 = tile_offsets<216>('s', 2);
    void (*kern)(uint32_t*, unsigned long long*, int, int, int, Off3, Off3) = k<V >= 3 ? 0 : V, NTH>;
    if (V == 3) kern = kp<NTH>;
    if (V == 4) kern = kd<NTH>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTH), lds, 0, d, c, 2, 0, ROUNDS, o0, o1);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTH), lds, 0, d, c, iters, 0, ROUNDS, o0, o1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), c, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double wslots = (double)iters * ROUNDS * (NTH / 64 / 4.0);     // wave-slots per SIMD
    printf("%-44s %4d thr  %8.3f ms   memtime %10.0f   -> %6.1f memtime ticks, %6.1f ns per wave-slot per SIMD\n", name, NTH, ms, mean,
           mean / wslots, ms * 1e6 / wslots);
}

int main() {
    uint32_t* d; unsigned long long* c;
    hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&c, 256 * 8);
    run<0, 1024>("slot as in the kernel", d, c);
    run<1, 1024>("no round / padding tests", d, c);
    run<2, 1024>("no LDS (loads -> register arithmetic)", d, c);
    run<8, 1024>("asm gathers, waits at 5 and 0", d, c);
    run<9, 1024>("asm gathers, one wait", d, c);
    run<1, 1024>("no round / padding tests (again)", d, c);
    run<5, 1024>("conflict-free gathers", d, c);
    run<6, 1024>("consecutive pixel addresses", d, c);
    run<7, 1024>("both", d, c);
    run<5, 512>("conflict-free gathers", d, c);
    run<7, 512>("both", d, c);
    run<3, 1024>("pipelined pixel loads", d, c);
    run<4, 1024>("deep pipeline", d, c);
    run<4, 512>("deep pipeline", d, c);
    run<4, 256>("deep pipeline", d, c);
    run<3, 512>("pipelined pixel loads", d, c);
    run<3, 256>("pipelined pixel loads", d, c);
    run<0, 512>("slot as in the kernel", d, c);
    run<1, 512>("no round / padding tests", d, c);
    run<2, 512>("no LDS", d, c);
    run<0, 256>("slot as in the kernel", d, c);
    run<1, 256>("no round / padding tests", d, c);
    run<2, 256>("no LDS", d, c);
    return 0;
}
