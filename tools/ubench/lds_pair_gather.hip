// Round 5, VERDICT r4 #2(a): can ONE LDS instruction fetch the two ADJACENT table entries every simplex walk touches (its
// stride-1 step), so that a walk costs 4 gathers instead of 5?  Rates of a CU's LDS for wave64 gathers at RANDOM addresses of
// a 134-KB table (one stage-2 piece), one 1024-thread workgroup per CU, ten instructions per wave and iteration, addresses
// formed once in front of the loop (as lerf_ubench_lds_gather does for bench.py's roofline_lds):
//   b32        ds_read_b32, dword aligned                           (today's stage-2 gather)
//   b64 al     ds_read_b64, 8-byte aligned
//   b64 odd    ds_read_b64 at 4 (mod 8)                             (a pair that straddles an 8-byte boundary)
//   b64 any    ds_read_b64 at any dword address                     (what the walk would issue)
//   read2      ds_read2_b32 offset0:0 offset1:1 at any dword address
//   u8         ds_read_u8 at any byte address                       (today's stage-1 gather)
//   u16 any    ds_read_u16 at any byte address                      (an adjacent byte pair of a stage-1 LUT)
//   u16 even   ds_read_u16 at even byte addresses
// and the correctness of the misaligned forms (values against a byte-wise read of the same LDS).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/lds_pair_gather.hip -o tools/ubench/lds_pair_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int TABLE = 7 * 4913 * 4;
constexpr int LDS = (TABLE + 64 + 15) / 16 * 16;
enum { B32, B64_AL, B64_ODD, B64_ANY, READ2, U8, U16_ANY, U16_EVEN, NMODES };
static const char* NAMES[NMODES] = {"ds_read_b32 (dword aligned)", "ds_read_b64 (8-byte aligned)", "ds_read_b64 at 4 mod 8", "ds_read_b64 at any dword",
                                    "ds_read2_b32 offset1:1 at any dword", "ds_read_u8 at any byte", "ds_read_u16 at any byte", "ds_read_u16 at even bytes"};

template <int MODE>
__global__ void __launch_bounds__(1024) gather(uint32_t* __restrict__ sink, uint32_t* __restrict__ bad, int iters, int conflict_free) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS / 4; i += 1024) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 3) * 40503u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    uint32_t a[10];
    uint32_t x = (uint32_t)tid * 747796405u + blockIdx.x * 2891336453u + 1u;
#pragma unroll
    for (int g = 0; g < 10; ++g) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = (x >> 8) % (uint32_t)(TABLE / 8 - 1);         // a random 8-byte slot
        uint32_t off;
        if (MODE == B32 || MODE == B64_ANY || MODE == READ2) off = r * 8u + ((x >> 4) & 1u) * 4u;
        else if (MODE == B64_AL) off = r * 8u;
        else if (MODE == B64_ODD) off = r * 8u + 4u;
        else if (MODE == U8 || MODE == U16_ANY) off = r * 8u + ((x >> 4) & 7u);
        else off = r * 8u + ((x >> 4) & 6u);
        if (conflict_free) {                                             // lane-linear rows: the ceiling of each form
            const uint32_t w = (MODE == B64_AL || MODE == B64_ODD || MODE == B64_ANY || MODE == READ2) ? 8u : 4u;
            off = (uint32_t)lane * w + (uint32_t)(g + 10 * (tid >> 6)) * 512u + (MODE == B64_ODD ? 4u : 0u);
        }
        a[g] = base + off;
    }
    // correctness of the first address of every lane
    {
        uint32_t lo = 0, hi = 0;
        const uint8_t* p = smem + (a[0] - base);
        uint32_t wl = 0, wh = 0;
        if (MODE == B32) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a[0])); for (int k = 3; k >= 0; --k) wl = (wl << 8) | p[k]; }
        else if (MODE == READ2) {
            uint64_t v; asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[0]));
            lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
            for (int k = 3; k >= 0; --k) wl = (wl << 8) | p[k];
            for (int k = 7; k >= 4; --k) wh = (wh << 8) | p[k];
        } else if (MODE == U8) { asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a[0])); wl = p[0]; }
        else if (MODE == U16_ANY || MODE == U16_EVEN) { asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(lo) : "v"(a[0])); wl = p[0] | (p[1] << 8); }
        else {
            uint64_t v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[0]));
            lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
            for (int k = 3; k >= 0; --k) wl = (wl << 8) | p[k];
            for (int k = 7; k >= 4; --k) wh = (wh << 8) | p[k];
        }
        if (lo != wl || hi != wh) atomicAdd(bad, 1u);
    }
    __syncthreads();
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == B32 || MODE == U8 || MODE == U16_ANY || MODE == U16_EVEN) {
            uint32_t d[10];
#pragma unroll
            for (int g = 0; g < 10; ++g) {
                if (MODE == B32) asm volatile("ds_read_b32 %0, %1" : "=v"(d[g]) : "v"(a[g]));
                else if (MODE == U8) asm volatile("ds_read_u8 %0, %1" : "=v"(d[g]) : "v"(a[g]));
                else asm volatile("ds_read_u16 %0, %1" : "=v"(d[g]) : "v"(a[g]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]));
#pragma unroll
            for (int g = 0; g < 10; ++g) acc += d[g];
        } else {
            uint64_t d[10];
#pragma unroll
            for (int g = 0; g < 10; ++g) {
                if (MODE == READ2) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(d[g]) : "v"(a[g]));
                else asm volatile("ds_read_b64 %0, %1" : "=v"(d[g]) : "v"(a[g]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]));
#pragma unroll
            for (int g = 0; g < 10; ++g) acc += (uint32_t)d[g] + (uint32_t)(d[g] >> 32);
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
static void run(uint32_t* sink, uint32_t* bad, int cus) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gather<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    for (int cf = 0; cf < 2; ++cf) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(gather<MODE>, dim3(cus), dim3(1024), LDS, 0, sink, bad, 10, cf);
        hipDeviceSynchronize();
        uint32_t nb = 0;
        hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
        float best = 1e9f;
        const int iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(gather<MODE>, dim3(cus), dim3(1024), LDS, 0, sink, bad, iters, cf);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("%-38s %-13s %7.3f ns per wave-instruction per CU   (%u wrong values of %d)\n", NAMES[MODE], cf ? "conflict-free" : "random", best * 1e6 / (iters * 10.0 * 16.0),
               nb, cus * 1024);
    }
}

int main() {
    uint32_t *sink, *bad;
    hipMalloc(&sink, 4); hipMalloc(&bad, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs; one 1024-thread workgroup per CU, 10 instructions per wave and iteration, 134-KB table\n", p.name, cus);
    run<B32>(sink, bad, cus); run<B64_AL>(sink, bad, cus); run<B64_ODD>(sink, bad, cus); run<B64_ANY>(sink, bad, cus); run<READ2>(sink, bad, cus);
    run<U8>(sink, bad, cus); run<U16_ANY>(sink, bad, cus); run<U16_EVEN>(sink, bad, cus);
    return 0;
}
