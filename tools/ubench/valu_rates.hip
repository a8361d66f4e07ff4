// Micro-benchmark: per-instruction VALU throughput on gfx950 (cycles per wave64 instruction per SIMD
// with 4 waves per SIMD, 8 independent chains per wave).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    uint32_t b = seed | 0x3f800001u, c = seed * 7 + 3;
    unsigned long long m64 = 0x5555555555555555ull * seed;
    for (int i = 0; i < iters; ++i) {
#define ONE(n)                                                                                                   \
        if (OP == 0) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                   \
        if (OP == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                   \
        if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                        \
        if (OP == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                   \
        if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                    \
        if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                   \
        if (OP == 6) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r##n) : "v"(b));                            \
        if (OP == 7) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(r##n) : "v"(b)); \
        if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                \
        if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                       \
        if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 11) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(r##n) : "v"(b));                               \
        if (OP == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(r##n));                                               \
        if (OP == 13) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r##n));                                        \
        if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(rr##n) : "v"(bb), "v"(cc));                \
        if (OP == 15) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                   \
        if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                      \
        if (OP == 17) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                      \
        if (OP == 18) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                    \
        if (OP == 19) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##n) : "v"(b));                              \
        if (OP == 20) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r##n) : "v"(b));                              \
        if (OP == 21) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r##n));                                        \
        if (OP == 22) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r##n));                                        \
        if (OP == 23) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                   \
        if (OP == 24) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 25) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(b) : "vcc");                  \
        if (OP == 26) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r##n), "v"(b) : "vcc");                      \
        if (OP == 27) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 28) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(r##n));                                         \
        if (OP == 29) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                          \
        if (OP == 30) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 31) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                    \
        if (OP == 32) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r##n) : "v"(b));                          \
        if (OP == 33) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                   \
        if (OP == 34) asm volatile("v_rcp_f32 %0, %0" : "+v"(r##n));                                               \
        if (OP == 35) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 36) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 37) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                  \
        if (OP == 38) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r##n));                                        \
        if (OP == 39) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r##n));                                        \
        if (OP == 40) asm volatile("v_mov_b32 %0, %1" : "+v"(r##n) : "v"(b));                                      \
        if (OP == 41) asm volatile("v_rndne_f32 %0, %0" : "+v"(r##n));                                             \
        if (OP == 42) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r##n));                                           \
        if (OP == 43) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(r##n) : "v"(b)); \
        if (OP == 44) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r##n) : "v"(b)); \
        if (OP == 45) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                      \
        if (OP == 46) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "s"(m64));             \
        if (OP == 47) asm volatile("v_cmp_gt_f32_e64 %0, |%1|, %2" : "=s"(m64) : "v"(r##n), "v"(b));               \
        if (OP == 48) asm volatile("v_max_f32 %0, |%0|, %1" : "+v"(r##n) : "v"(b));                                \
        if (OP == 49) asm volatile("v_cmp_gt_f32_e64 %1, |%0|, %2\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "+v"(r##n), "=s"(m64) : "v"(b)); \
        if (OP == 50) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r##n) : "v"(b));                          \
        if (OP == 51) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(rr##n) : "v"(b), "v"(c), "s"(m64));
        uint64_t rr0 = r0, rr1 = r1, rr2 = r2, rr3 = r3, rr4 = r4, rr5 = r5, rr6 = r6, rr7 = r7, bb = b, cc = c;
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
        if (OP == 14 || OP == 51) { r0 = (uint32_t)rr0; r1 = (uint32_t)rr1; r2 = (uint32_t)rr2; r3 = (uint32_t)rr3; r4 = (uint32_t)rr4; r5 = (uint32_t)rr5; r6 = (uint32_t)rr6; r7 = (uint32_t)rr7; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    const int iters = 4000, blocks = 512;      // 2 blocks of 1024 per CU -> 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * 32 * (blocks * 16.0 / (256 * 4));     // wave-instructions per SIMD
    printf("%-22s %8.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    uint32_t* d; hipMalloc(&d, 512 * 1024 * 4);
    run<2>("v_fma_f32", d); run<10>("v_mul_f32", d); run<1>("v_max_f32", d); run<14>("v_pk_fma_f32", d);
    run<0>("v_max_u32", d); run<3>("v_and_b32", d); run<5>("v_add_u32", d); run<4>("v_mad_u32_u24", d);
    run<15>("v_mad_i32_i24", d); run<6>("v_lshl_or_b32", d); run<9>("v_add3_u32", d); run<7>("v_sub_u32_sdwa", d);
    run<8>("v_mul_lo_u32", d); run<11>("v_pk_max_u16", d); run<16>("v_perm_b32", d); run<17>("v_med3_u32", d);
    run<18>("v_max3_u32", d); run<13>("v_cvt_f32_ubyte1", d); run<12>("v_exp_f32", d);
    run<19>("v_mul_u32_u24", d); run<20>("v_mul_i32_i24", d); run<21>("v_lshlrev_b32", d); run<22>("v_lshrrev_b32", d);
    run<23>("v_or_b32", d); run<24>("v_sub_u32", d); run<25>("v_cndmask_b32", d); run<26>("v_cmp_gt_u32", d);
    run<27>("v_min_u32", d); run<28>("v_bfe_u32", d); run<29>("v_fmac_f32", d); run<30>("v_xor_b32", d);
    run<31>("v_and_or_b32", d); run<32>("v_lshl_add_u32", d); run<33>("v_dot4_u32_u8", d); run<34>("v_rcp_f32", d);
    run<35>("v_add_f32", d); run<36>("v_sub_f32", d); run<37>("v_min_f32", d); run<38>("v_cvt_f32_ubyte0", d);
    run<39>("v_ashrrev_i32", d); run<40>("v_mov_b32", d); run<41>("v_rndne_f32", d); run<42>("v_cvt_u32_f32", d);
    run<46>("v_cndmask_b32_e64 sgpr mask", d); run<47>("v_cmp_gt_f32_e64 -> sgpr", d); run<48>("v_max_f32 |abs|", d); run<49>("v_cmp + v_cndmask pair (2 instr)", d); run<50>("v_cvt_pk_u8_f32", d); run<51>("v_mad_u64_u32", d);
    run<43>("v_add_u32_sdwa", d); run<44>("v_mul_u32_u24_sdwa", d); run<45>("v_min3_f32", d);
    return 0;
}
