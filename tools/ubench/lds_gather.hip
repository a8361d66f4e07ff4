// Micro-benchmarks behind the round-3 design decisions (run on the GPU box):
//   (1) CLOCKS: s_memtime (shader clock counter) against s_memrealtime (100 MHz) and the host's event clock, for one
//       workgroup alone and for 256 x 1024 threads of the stage-2 slot replay -- which frequency does the chip run
//       this code at, and which counter do the stamps / PMC summaries count in?
//   (2) UNALIGNED LDS dword gathers: does ds_read_b32 at a byte address 3i return the four bytes at 3i..3i+3 on
//       gfx950, and what does it cost against the aligned gather?  (3-byte stage-2 LUT entries would shrink a piece
//       by a quarter: two bins instead of three, or two resident pieces.)
//   (3) the stage-2 slot with 3-byte entries: Walk<3>, MACs as mad24(w, d) [e0 | e1 << 8 | e2 << 16 summed unmasked]
//       + bfe + mad24 for e1 (the e0 / e2 sums come out as full - (e1sum << 8)).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I lerf-pytorch_amd/csrc tools/ubench/lds_gather.hip -o tools/ubench/lds_gather
#include "../../lerf-pytorch_amd/csrc/lerf_fused.hip"
#include <cstdio>
#include <vector>

using namespace lerf;
using namespace lerf::fused;

__device__ __forceinline__ unsigned long long realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

constexpr int ROUNDS = 14;
constexpr int LDS_TEST = 16384 + PIECE_LDS + 4096;

// ---- (2) raw gathers: G gathers per iteration at addresses base + SCALE * idx, idx random per lane and gather
template <int SCALE, int NTH>
__global__ void __launch_bounds__(1024) gather_kernel(uint32_t* out, unsigned long long* cyc, int iters, uint32_t* bad) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < LDS_TEST / 4; i += NTH) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t base = lds_addr(smem + 16384);
    constexpr uint32_t NENT = 7 * 4913;
    uint32_t x = (uint32_t)tid * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
    // correctness of the unaligned read on a few addresses per thread
    if (bad != nullptr) {
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = base + (x >> 8) % NENT * SCALE;
            uint32_t d;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a));
            const uint8_t* p = smem + (a - lds_addr(smem));
            const uint32_t want = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            if (d != want) atomicAdd(bad, 1u);
        }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint32_t a[10], d[10];
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            x = x * 1664525u + 1013904223u;
            a[g] = base + __umul24((x >> 10) % NENT, SCALE);
        }
#pragma unroll
        for (int g = 0; g < 10; ++g) asm volatile("ds_read_b32 %0, %1" : "=v"(d[g]) : "v"(a[g]));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]));
#pragma unroll
        for (int g = 0; g < 10; ++g) acc += d[g];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NTH + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- (3) the slot, SCALE = 4 (as in the kernel) or 3 (packed 3-byte entries)
template <int SCALE>
__device__ __forceinline__ void slot(uint32_t cpa, const Off3& o0, const Off3& o1, uint32_t qbase, uint32_t& accA, uint32_t& accB) {
    const unsigned st_a = kStrideA * SCALE, st_b = kStrideB * SCALE, st_c = kStrideC * SCALE, st_d = kStrideD * SCALE;
    uint32_t ra = lds_pixel_hi(cpa);
    uint32_t rb0 = lds_pixel_hi(cpa + (uint32_t)o0.o[0]), rc0 = lds_pixel_hi(cpa + (uint32_t)o0.o[1]), rd0 = lds_pixel_hi(cpa + (uint32_t)o0.o[2]);
    uint32_t rb1 = lds_pixel_hi(cpa + (uint32_t)o1.o[0]), rc1 = lds_pixel_hi(cpa + (uint32_t)o1.o[1]), rd1 = lds_pixel_hi(cpa + (uint32_t)o1.o[2]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra), "+v"(rb0), "+v"(rc0), "+v"(rd0), "+v"(rb1), "+v"(rc1), "+v"(rd1));
    const int basea = (int)(__umul24(msb_of(ra) & 3u, kStrideA * SCALE) + qbase);
    const unsigned ka = key_of(ra, st_a);
    const Walk<SCALE> W0 = simplex_walk<SCALE>(ka, basea, rb0, rc0, rd0, st_b, st_c, st_d);
    const Walk<SCALE> W1 = simplex_walk<SCALE>(ka, basea, rb1, rc1, rd1, st_b, st_c, st_d);
    uint32_t d0[5], d1[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) d0[n] = W0.ld32(n);
#pragma unroll
    for (int n = 0; n < 5; ++n) d1[n] = W1.ld32(n);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned w0[5] = {(unsigned)kQ - W0.f0, W0.f0 - W0.f1, W0.f1 - W0.f2, W0.f2 - W0.f3, W0.f3};
    const unsigned w1[5] = {(unsigned)kQ - W1.f0, W1.f0 - W1.f1, W1.f1 - W1.f2, W1.f2 - W1.f3, W1.f3};
    uint32_t a = accA, bb = accB;
    if (SCALE == 4) {
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w0[n], d0[n]); bb += __umul24(w0[n], d0[n] >> 24); }
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w1[n], d1[n]); bb += __umul24(w1[n], d1[n] >> 24); }
    } else {
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w0[n], d0[n]); bb += __umul24(w0[n], __builtin_amdgcn_ubfe(d0[n], 8, 8)); }
#pragma unroll
        for (int n = 0; n < 5; ++n) { a += __umul24(w1[n], d1[n]); bb += __umul24(w1[n], __builtin_amdgcn_ubfe(d1[n], 8, 8)); }
    }
    accA = a; accB = bb;
}

template <int SCALE, int NTH>
__global__ void __launch_bounds__(1024) slot_kernel(uint32_t* out, unsigned long long* cyc, int iters, Off3 o0, Off3 o1) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < LDS_TEST / 4; i += NTH) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t bt_a0 = lds_addr(smem), qbase = lds_addr(smem + 16384);
    uint32_t sa[ROUNDS], accA[ROUNDS], accB[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        sa[r] = 4 * 216 + ((tid * 7 + r * 1031 + blockIdx.x * 13) % (64 * 216));
        accA[r] = 0; accB[r] = 0;
    }
    __syncthreads();
    const unsigned long long r0 = realtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint32_t bt_a = bt_a0 + (uint32_t)(it & 1) * 3u;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) slot<SCALE>(bt_a + sa[r], o0, o1, qbase, accA[r], accB[r]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = realtime();
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) x ^= accA[r] + accB[r];
    out[blockIdx.x * NTH + tid] = x;
    if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[1024 + blockIdx.x] = r1 - r0; }
}

template <typename K, typename... A>
static float timed(K kern, int blocks, int nth, A... args) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TEST);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nth), LDS_TEST, 0, args...);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    uint32_t* d; unsigned long long* c; uint32_t* bad;
    hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&c, 2048 * 8); hipMalloc(&bad, 4);
    std::vector<unsigned long long> h(2048);
    Off3 o0 = tile_offsets<216>('s', 0), o1 = tile_offsets<216>('s', 2);

    // (1) clocks
    for (int blocks : {1, 256}) {
        timed(slot_kernel<4, 1024>, blocks, 1024, d, c, 20, o0, o1);
        const int iters = 2000;
        const float ms = timed(slot_kernel<4, 1024>, blocks, 1024, d, c, iters, o0, o1);
        hipMemcpy(h.data(), c, 2048 * 8, hipMemcpyDeviceToHost);
        double mt = 0, rt = 0;
        for (int i = 0; i < blocks; ++i) { mt += (double)h[i]; rt += (double)h[1024 + i]; }
        mt /= blocks; rt /= blocks;
        printf("CLOCK %3d workgroups x 1024: event %.3f ms | s_memrealtime %.0f ticks = %.3f ms at 100 MHz | s_memtime %.0f ticks = %.1f MHz against s_memrealtime, %.1f MHz against the event time\n",
               blocks, ms, rt, rt / 1e5, mt, mt / (rt / 100.0), mt / (ms * 1e3));
    }
    // (2) raw gathers
    hipMemset(bad, 0, 4);
    timed(gather_kernel<3, 1024>, 256, 1024, d, c, 1, bad);
    uint32_t nbad = 0; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
    printf("UNALIGNED ds_read_b32 at 3i: %u wrong of %d\n", nbad, 256 * 1024 * 8);
    hipMemset(bad, 0, 4);
    timed(gather_kernel<4, 1024>, 256, 1024, d, c, 1, bad);
    hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
    printf("aligned   ds_read_b32 at 4i: %u wrong\n", nbad);
    {
        const int iters = 2000;
        const float m4 = timed(gather_kernel<4, 1024>, 256, 1024, d, c, iters, (uint32_t*)nullptr);
        const float m3 = timed(gather_kernel<3, 1024>, 256, 1024, d, c, iters, (uint32_t*)nullptr);
        const float m4h = timed(gather_kernel<4, 512>, 256, 512, d, c, iters, (uint32_t*)nullptr);
        const float m3h = timed(gather_kernel<3, 512>, 256, 512, d, c, iters, (uint32_t*)nullptr);
        const double wg = (double)iters * 10 * 16;        // wave-gathers per CU
        printf("GATHER 10 random dword gathers per iteration, 16 waves/CU: aligned %.3f ms = %.2f ns per wave-gather per CU; 3-byte stride %.3f ms = %.2f ns (x%.2f)\n",
               m4, m4 * 1e6 / wg, m3, m3 * 1e6 / wg, m3 / m4);
        printf("GATHER  8 waves/CU: aligned %.3f ms = %.2f ns; 3-byte stride %.3f ms = %.2f ns (x%.2f)\n", m4h, m4h * 1e6 / (wg / 2), m3h,
               m3h * 1e6 / (wg / 2), m3h / m4h);
    }
    // (3) the slot
    {
        const int iters = 400;
        timed(slot_kernel<4, 1024>, 256, 1024, d, c, 20, o0, o1);
        for (int rep = 0; rep < 2; ++rep) {
            const float a4 = timed(slot_kernel<4, 1024>, 256, 1024, d, c, iters, o0, o1);
            const float a3 = timed(slot_kernel<3, 1024>, 256, 1024, d, c, iters, o0, o1);
            const double ws = (double)iters * ROUNDS * 4;  // wave-slots per SIMD
            printf("SLOT 1024 thr: dword entries %.3f ms = %.1f ns per wave-slot per SIMD; 3-byte entries %.3f ms = %.1f ns (x%.3f)\n", a4,
                   a4 * 1e6 / ws, a3, a3 * 1e6 / ws, a3 / a4);
        }
        const float b4 = timed(slot_kernel<4, 512>, 256, 512, d, c, iters, o0, o1);
        const float b3 = timed(slot_kernel<3, 512>, 256, 512, d, c, iters, o0, o1);
        const double ws = (double)iters * ROUNDS * 2;
        printf("SLOT  512 thr: dword entries %.3f ms = %.1f ns; 3-byte entries %.3f ms = %.1f ns (x%.3f)\n", b4, b4 * 1e6 / ws, b3, b3 * 1e6 / ws,
               b3 / b4);
    }
    return 0;
}
