// Micro-benchmark, second set: issue cost of the mixed-precision / packed / conversion instructions that stage 3 could use
// instead of v_cvt_f32_ubyte + v_mul/v_fma (cycles per wave64 instruction per SIMD, 8 waves per SIMD, 8 independent chains
// per wave -- same harness as valu_rates.hip).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    uint32_t b = seed | 0x3f800001u, c = seed * 7 + 3;
    for (int i = 0; i < iters; ++i) {
#define ONE(n)                                                                                                   \
        if (OP == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r##n) : "v"(b), "v"(c)); \
        if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r##n) : "v"(b), "v"(c)); \
        if (OP == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(r##n) : "v"(b), "v"(c)); \
        if (OP == 3) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r##n));                                            \
        if (OP == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r##n)); \
        if (OP == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x40000000" : "+v"(r##n) : "v"(b));                     \
        if (OP == 6) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r##n));                                            \
        if (OP == 7) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r##n) : "v"(b), "v"(c));                  \
        if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(rr##n) : "v"(bb));                              \
        if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(rr##n) : "v"(bb));                              \
        if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                  \
        if (OP == 11) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                       \
        if (OP == 12) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                    \
        if (OP == 13) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r##n) : "v"(b));                               \
        if (OP == 14) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n) : "v"(b)); \
        if (OP == 15) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n) : "v"(b)); \
        if (OP == 16) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r##n) : "v"(b));                                \
        if (OP == 17) asm volatile("v_fma_f32 %0, %0, %1, 2.0" : "+v"(r##n) : "v"(b));                             \
        if (OP == 18) asm volatile("v_fma_f32 %0, -%0, %1, 1.0" : "+v"(r##n) : "v"(b));                            \
        if (OP == 19) asm volatile("v_exp_f32 %0, -%0" : "+v"(r##n));                                              \
        if (OP == 20) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##n) : "s"(seed));                               \
        if (OP == 21) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(r##n) : "v"(b));                                \
        if (OP == 22) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                    \
        if (OP == 23) asm volatile("v_cvt_f16_u16 %0, %0" : "+v"(r##n));                                           \
        if (OP == 24) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                   \
        if (OP == 25) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c));                       \
        if (OP == 26) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(r##n) : "v"(b));                          \
        if (OP == 27) asm volatile("v_sub_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2" : "+v"(r##n) : "v"(b), "v"(c));   \
        if (OP == 28) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c)); \
        if (OP == 29) asm volatile("v_cvt_f32_ubyte2 %0, %0\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(b), "v"(c)); \
        if (OP == 30) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(r##n), "+v"(q##n) : "v"(b), "v"(c)); \
        if (OP == 31) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(r##n), "+v"(q##n) : "v"(b), "v"(c)); \
        if (OP == 32) asm volatile("v_cvt_f32_ubyte1 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(r##n), "+v"(q##n) : "v"(b), "v"(c)); \
        if (OP == 33) asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(r##n), "+v"(q##n) : "v"(b), "v"(c)); \
        if (OP == 34) asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n\tv_add_u32 %1, %1, %2" : "+v"(r##n), "+v"(q##n) : "v"(b), "v"(c));
        uint64_t rr0 = r0, rr1 = r1, rr2 = r2, rr3 = r3, rr4 = r4, rr5 = r5, rr6 = r6, rr7 = r7, bb = b;
        uint32_t q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
        if (OP == 8 || OP == 9) { r0 = (uint32_t)rr0; r1 = (uint32_t)rr1; r2 = (uint32_t)rr2; r3 = (uint32_t)rr3; r4 = (uint32_t)rr4; r5 = (uint32_t)rr5; r6 = (uint32_t)rr6; r7 = (uint32_t)rr7; }
        if (OP >= 30) { r0 ^= q0; r1 ^= q1; r2 ^= q2; r3 ^= q3; r4 ^= q4; r5 ^= q5; r6 ^= q6; r7 ^= q7; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int OP>
void run(const char* name, uint32_t* d, int per = 1) {
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
    double instr_per_simd = (double)iters * 32 * (blocks * 16.0 / (256 * 4));     // asm groups per SIMD
    printf("%-44s %8.3f ms  -> %.2f ns per group of %d per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / instr_per_simd, per, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    uint32_t* d; hipMalloc(&d, 512 * 1024 * 4);
    run<2>("v_fma_mix_f32 (all f32)", d); run<1>("v_fma_mix_f32 (src0 f16 lo)", d); run<0>("v_fma_mix_f32 (src0 f16 hi)", d);
    run<3>("v_cvt_f32_f16", d); run<4>("v_cvt_f32_f16_sdwa WORD_1", d); run<5>("v_fmaak_f32", d); run<6>("v_cvt_f32_u32", d);
    run<7>("v_dot2_f32_f16", d); run<8>("v_pk_mul_f32", d); run<9>("v_pk_add_f32", d); run<10>("v_alignbit_b32", d);
    run<11>("v_bfi_b32", d); run<12>("v_pk_mad_u16", d); run<13>("v_pk_add_u16", d); run<14>("v_mov_b32_dpp", d);
    run<15>("v_add_u32_dpp", d); run<16>("v_ldexp_f32", d); run<17>("v_fma_f32 inline const", d);
    run<18>("v_fma_f32 neg + const", d); run<19>("v_exp_f32 neg", d); run<20>("v_mul_f32 sgpr", d); run<21>("v_add_f32 |abs|", d);
    run<22>("v_pk_fma_f16", d); run<23>("v_cvt_f16_u16", d); run<24>("v_mad_u32_u16", d); run<25>("v_xad_u32", d);
    run<26>("v_add_lshl_u32", d);
    run<27>("sub_f32 + mul_f32 dependent", d, 2); run<28>("fma_mix(f16) + fma_f32 dependent", d, 2);
    run<29>("cvt_ubyte2 + fma_f32 dependent", d, 2);
    run<30>("exp + 1 independent fma", d, 2); run<31>("exp + 3 independent fma", d, 4);
    run<32>("cvt_ubyte1 + 1 independent fma", d, 2); run<33>("mad24 + 1 independent fma", d, 2);
    run<34>("mad24 + 1 independent add_u32", d, 2);
    return 0;
}
