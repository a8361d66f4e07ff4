// Is v_cndmask_b32 reading VCC (the VOP2 form hipcc emits for a select) slower than the VOP3 form with an SGPR-pair mask?
// 4 waves per SIMD, 8 independent chains per wave, compare + select pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    uint32_t b = seed * 2654435761u, c = seed * 7 + 3;
    unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
#define ONE(n)                                                                                                                   \
        if (OP == 0) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r##n) : "v"(b), "v"(c) : "vcc"); \
        if (OP == 1) asm volatile("v_cmp_gt_u32_e64 %3, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, %3" : "+v"(r##n) : "v"(b), "v"(c), "s"(m)); \
        if (OP == 2) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(c));                                       \
        if (OP == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##n) : "v"(c), "s"(m));                                \
        if (OP == 4) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r##n), "v"(b) : "vcc");                                        \
        if (OP == 5) asm volatile("v_min_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %2" : "+v"(r##n) : "v"(b), "v"(c));                   \
        if (OP == 6) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %2, %0, vcc\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %2, %0, vcc" : "+v"(r##n) : "v"(b), "v"(c) : "vcc"); \
        if (OP == 7) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %0, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r##n) : "v"(b), "v"(c) : "vcc"); \
        if (OP == 8) asm volatile("s_and_b64 vcc, exec, %3\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r##n) : "v"(b), "v"(c), "s"(m) : "vcc", "scc");
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}
template <int OP>
void run(const char* name, uint32_t* d, int per) {
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, iters, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 32 * per * 4;      // wave-instructions per SIMD (4 waves per SIMD)
    printf("%-52s %8.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / n);
    fflush(stdout);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_cmp (vcc) + v_cndmask_b32_e32 (vcc)", d, 2);
    run<1>("v_cmp_e64 (sgpr pair) + v_cndmask_b32_e64", d, 2);
    run<2>("v_cndmask_b32_e32 alone (vcc never written)", d, 1);
    run<3>("v_cndmask_b32_e64 alone", d, 1);
    run<4>("v_cmp (vcc) alone", d, 1);
    run<5>("v_min_u32 + v_max_u32 (a clamp without selects)", d, 2);
    run<6>("v_cmp (vcc) + 4 x v_cndmask_b32_e32", d, 5);
    run<7>("v_cmp (vcc) + 3 v_add + v_cndmask_b32_e32", d, 5);
    run<8>("s_and_b64 vcc + v_cndmask_b32_e32 (per VALU instr)", d, 1);
    return 0;
}
