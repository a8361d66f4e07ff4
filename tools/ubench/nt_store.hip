// WRITE_SIZE calibration: 199 MB written once with fully coalesced dword stores, nontemporal and regular.
// rocprofv3 --pmc WRITE_SIZE -- tools/ubench/nt_store
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
int main() {
    const size_t bytes = 8ull * 2160 * 3840 * 3, n = bytes / 4;
    uint32_t* d; hipMalloc(&d, bytes);
    hipLaunchKernelGGL(fill<true>, dim3(4096), dim3(256), 0, 0, d, n);
    hipLaunchKernelGGL(fill<false>, dim3(4096), dim3(256), 0, 0, d, n);
    hipDeviceSynchronize();
    printf("wrote %zu bytes twice\n", bytes);
    return 0;
}
