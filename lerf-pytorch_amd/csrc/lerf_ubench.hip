// Calibration kernels behind bench.py's roofline_lds: what a CU's LDS serves in wave64 dword gathers per second, measured
// in the run on the chip the bench runs on (round 3 hard-coded 3.53 ns from tools/ubench/lds_gather.hip).
//   pattern 0: every lane reads 10 RANDOM dwords of a 134-KB table per iteration (the stage-2 piece's size): 32 lanes of a
//              lane group fall on 32 banks like a hash, the busiest bank holds ~3.5 of them (MI355X_MICROARCH.md, LDS)
//   pattern 1: lane L reads dword L of ten different 256-byte rows: conflict-free, 2 LDS cycles per wave-instruction --
//              the ceiling no data-dependent gather reaches
// The addresses are formed once, in front of the loop; an iteration is ten ds_read_b32 and ten adds, the LDS is the only
// unit under load.  One 1024-thread workgroup per CU (the kernels' own shape: 16 waves).
#include "lerf_kernels.h"

namespace lerf {

constexpr int UB_TABLE = 7 * 4913 * 4;       // bytes: one stage-2 piece
constexpr int UB_LDS = (UB_TABLE + 15) / 16 * 16;

template <int PATTERN>
__global__ void __launch_bounds__(1024) ub_lds_gather_kernel(uint32_t* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ub_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < UB_LDS / 4; i += 1024) reinterpret_cast<uint32_t*>(ub_smem)[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)ub_smem;
    uint32_t a[10];
    uint32_t x = (uint32_t)tid * 747796405u + blockIdx.x * 2891336453u + 1u;
#pragma unroll
    for (int g = 0; g < 10; ++g) {
        x = x * 1664525u + 1013904223u;
        a[g] = PATTERN == 0 ? base + ((x >> 8) % (uint32_t)(UB_TABLE / 4)) * 4u
                            : base + (uint32_t)lane * 4u + (uint32_t)(g + 10 * (tid >> 6)) * 256u;
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t d[10];
#pragma unroll
        for (int g = 0; g < 10; ++g) asm volatile("ds_read_b32 %0, %1" : "=v"(d[g]) : "v"(a[g]));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]));
#pragma unroll
        for (int g = 0; g < 10; ++g) acc += d[g];
    }
    if (acc == 0x12345u) sink[0] = acc;          // keeps the loop; practically never taken
}

int launch_ubench_lds_gather(int pattern, int iters, int blocks, uint32_t* sink, hipStream_t st) {
    if (pattern == 0) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ub_lds_gather_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, UB_LDS) != hipSuccess) return LERF_ELAUNCH;
        hipLaunchKernelGGL(ub_lds_gather_kernel<0>, dim3(blocks), dim3(1024), UB_LDS, st, sink, iters);
    } else {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ub_lds_gather_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, UB_LDS) != hipSuccess) return LERF_ELAUNCH;
        hipLaunchKernelGGL(ub_lds_gather_kernel<1>, dim3(blocks), dim3(1024), UB_LDS, st, sink, iters);
    }
    return LERF_OK;
}

}  // namespace lerf
