// WRITE_SIZE calibration (rocprofv3 --pmc WRITE_SIZE -- tools/ubench/nt_store): 199 MB written once
//   fill<nt>, fill<regular>: fully coalesced dword stores, thread i -> dword i
//   tiles<ORDER>: the stage-3 store pattern of the tile kernel -- 4080 workgroups of 1024 threads, each a 128-row x 384-byte
//   block of a 3840x2160x3 frame; ORDER 0: task = (row pair, dword column), a lane stores its dword in both rows (the kernel's
//   order); ORDER 1: one row after the other (task = (row, dword column))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <bool NT>
__global__ void fill(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store((uint32_t)i, p + i); else p[i] = (uint32_t)i;
    }
}
template <int ORDER>
__global__ void __launch_bounds__(1024) tiles(uint8_t* out) {
    const int frame = blockIdx.x / 510, b = blockIdx.x % 510, ty = b / 30, tx = b % 30;
    const int rows = ty == 16 ? 2160 - 16 * 128 : 128;
    const size_t pitch = 3840 * 3;
    uint8_t* seg = out + (size_t)frame * 2160 * pitch + (size_t)ty * 128 * pitch + (size_t)tx * 384;
    if (ORDER == 0) {
        for (int t = threadIdx.x; t < (rows / 2) * 96; t += 1024) {
            const int g = t / 96, dw = t - g * 96;
            __builtin_nontemporal_store((uint32_t)t, reinterpret_cast<uint32_t*>(seg + (size_t)(2 * g) * pitch + dw * 4));
            __builtin_nontemporal_store((uint32_t)t, reinterpret_cast<uint32_t*>(seg + (size_t)(2 * g + 1) * pitch + dw * 4));
        }
    } else {
        for (int t = threadIdx.x; t < rows * 96; t += 1024) {
            const int r = t / 96, dw = t - r * 96;
            __builtin_nontemporal_store((uint32_t)t, reinterpret_cast<uint32_t*>(seg + (size_t)r * pitch + dw * 4));
        }
    }
}
int main() {
    const size_t bytes = 8ull * 2160 * 3840 * 3, n = bytes / 4;
    uint32_t* d; hipMalloc(&d, bytes);
    hipLaunchKernelGGL(fill<true>, dim3(4096), dim3(256), 0, 0, d, n);
    hipLaunchKernelGGL(fill<false>, dim3(4096), dim3(256), 0, 0, d, n);
    hipLaunchKernelGGL(tiles<0>, dim3(4080), dim3(1024), 0, 0, (uint8_t*)d);
    hipLaunchKernelGGL(tiles<1>, dim3(4080), dim3(1024), 0, 0, (uint8_t*)d);
    hipDeviceSynchronize();
    printf("wrote %zu bytes four times\n", bytes);
    return 0;
}
