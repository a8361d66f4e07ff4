// Micro-benchmark for north_star's "wavefront shuffles for the per-pixel weight normalisation", 4x4 support:
//   A  thread-local: one lane evaluates the 16 taps of one output pixel-channel and normalises in registers
//      (what every production kernel does);
//   B  cross-lane:   16 lanes per output pixel-channel, one tap each; the weight sum and the weighted sum are
//      reduced with a 4-step DPP butterfly inside the 16-lane row (quad_perm x2, row_ror:4, row_ror:8).
// Same arithmetic (pre-scaled steering-Gaussian form, exp2), same inputs (uint8 feat + 3 uint8 hyper planes, HWC),
// x2 SR geometry of a 1080p frame.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/s4_norm tools/ubench/s4_norm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int S = 4, C = 3;

__device__ __forceinline__ float tap_form(uint32_t d, float dx, float dy) {     // d = k0 | k1<<8 | k2<<16 | v<<24
    const float k0 = (float)(d & 0xFFu), k1 = (float)((d >> 8) & 0xFFu), k2 = (float)((d >> 16) & 0xFFu);
    const float m2rho = fmaf(k0, -4.0f / 255.0f, 2.0f);
    const float tx = k1 * dx, ty = k2 * dy;
    return fmaf(tx, m2rho * ty, fmaf(tx, tx, ty * ty));
}

__global__ void __launch_bounds__(256)
thread_local_kernel(const uint32_t* __restrict__ pk, int H, int W, int oH, int oW, const int* __restrict__ lr, const float* __restrict__ dr,
                    const int* __restrict__ lc, const float* __restrict__ dc, uint8_t* __restrict__ out) {
    const int xc = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (xc >= oW * C) return;
    const int j = xc / C, c = xc - j * C;
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const int rr = lr[i] + b, cc = lc[j] + a;
            const int rcl = min(max(rr, 0), H - 1), ccl = min(max(cc, 0), W - 1);
            const uint32_t d = pk[((size_t)rcl * W + ccl) * C + c];
            const float v = (rr == rcl && cc == ccl) ? (float)(d >> 24) : 0.0f;
            const float w = __builtin_amdgcn_exp2f(-tap_form(d, dr[i * S + b], dc[j * S + a]));
            num = fmaf(w, v, num);
            den += w;
        }
    float r = __builtin_amdgcn_rcpf(den);
    r = fmaf(fmaf(-den, r, 1.0f), r, r);
    out[((size_t)i * oW + j) * C + c] = (uint8_t)fminf(fmaxf(rintf(num * r), 0.0f), 255.0f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return x + __int_as_float(y);
}

__global__ void __launch_bounds__(256)
cross_lane_kernel(const uint32_t* __restrict__ pk, int H, int W, int oH, int oW, const int* __restrict__ lr, const float* __restrict__ dr,
                  const int* __restrict__ lc, const float* __restrict__ dc, uint8_t* __restrict__ out) {
    const int t = threadIdx.x & 15;                                  // tap of this lane: a = t / 4 (column), b = t % 4 (row)
    const int xc = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4), i = blockIdx.y;
    const bool live = xc < oW * C;
    const int xcc = live ? xc : oW * C - 1;
    const int j = xcc / C, c = xcc - j * C, a = t >> 2, b = t & 3;
    const int rr = lr[i] + b, cc = lc[j] + a;
    const int rcl = min(max(rr, 0), H - 1), ccl = min(max(cc, 0), W - 1);
    const uint32_t d = pk[((size_t)rcl * W + ccl) * C + c];
    const float v = (rr == rcl && cc == ccl) ? (float)(d >> 24) : 0.0f;
    const float w = __builtin_amdgcn_exp2f(-tap_form(d, dr[i * S + b], dc[j * S + a]));
    float num = w * v, den = w;
    num = dpp_add<0xB1>(num);  den = dpp_add<0xB1>(den);             // quad_perm [1,0,3,2]
    num = dpp_add<0x4E>(num);  den = dpp_add<0x4E>(den);             // quad_perm [2,3,0,1]
    num = dpp_add<0x124>(num); den = dpp_add<0x124>(den);            // row_ror:4
    num = dpp_add<0x128>(num); den = dpp_add<0x128>(den);            // row_ror:8
    if (t == 0 && live) {
        float r = __builtin_amdgcn_rcpf(den);
        r = fmaf(fmaf(-den, r, 1.0f), r, r);
        out[((size_t)i * oW + j) * C + c] = (uint8_t)fminf(fmaxf(rintf(num * r), 0.0f), 255.0f);
    }
}

static void axis(int n_in, int n_out, double s, std::vector<int>& left, std::vector<float>& dis, float scale) {
    left.resize(n_out); dis.resize((size_t)n_out * S);
    int pad = 0;
    for (int i = 0; i < n_out; ++i) {
        double g = i / s + (n_in - 1) / 2.0 - (n_out - 1) / (2 * s);
        int l = (int)std::ceil(g - S / 2.0 - 1.1920928955078125e-07);
        if (i == 0) pad = -l;
        left[i] = l;
        for (int k = 0; k < S; ++k) dis[(size_t)i * S + k] = (float)(((g + pad) - (l + pad + k)) * scale);
    }
}

int main() {
    const int H = 1080, W = 1920, oH = 2160, oW = 3840;
    const float gs = (10.0f / 255.0f) * 0.84932180028801904272f;      // (max_sigma/255) sqrt(0.5 log2 e)
    std::vector<uint32_t> pk((size_t)H * W * C);
    uint32_t st = 12345u;
    for (auto& x : pk) { st = st * 1664525u + 1013904223u; x = st; }
    std::vector<int> lr, lc; std::vector<float> dr, dc;
    axis(H, oH, 2.0, lr, dr, gs); axis(W, oW, 2.0, lc, dc, gs);
    uint32_t* dpk; int *dlr, *dlc; float *ddr, *ddc; uint8_t *oa, *ob;
    hipMalloc(&dpk, pk.size() * 4); hipMalloc(&dlr, lr.size() * 4); hipMalloc(&dlc, lc.size() * 4);
    hipMalloc(&ddr, dr.size() * 4); hipMalloc(&ddc, dc.size() * 4);
    hipMalloc(&oa, (size_t)oH * oW * C); hipMalloc(&ob, (size_t)oH * oW * C);
    hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dlr, lr.data(), lr.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dlc, lc.data(), lc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddr, dr.data(), dr.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ddc, dc.data(), dc.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 ga((oW * C + 255) / 256, oH), gb((oW * C + 15) / 16, oH);
    float ms[2] = {0, 0};
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 12; ++rep) {
            if (rep == 2) hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(thread_local_kernel, ga, dim3(256), 0, 0, dpk, H, W, oH, oW, dlr, ddr, dlc, ddc, oa);
            else hipLaunchKernelGGL(cross_lane_kernel, gb, dim3(256), 0, 0, dpk, H, W, oH, oW, dlr, ddr, dlc, ddc, ob);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[variant], e0, e1);
        ms[variant] /= 10;
    }
    std::vector<uint8_t> ha((size_t)oH * oW * C), hb(ha.size());
    hipMemcpy(ha.data(), oa, ha.size(), hipMemcpyDeviceToHost); hipMemcpy(hb.data(), ob, hb.size(), hipMemcpyDeviceToHost);
    size_t diff = 0; int maxd = 0;
    for (size_t k = 0; k < ha.size(); ++k) { int d = std::abs((int)ha[k] - (int)hb[k]); diff += d != 0; maxd = d > maxd ? d : maxd; }
    const double mpix = (double)oH * oW / 1e6;
    printf("4x4 steering-Gaussian resampling, 1920x1080 -> 3840x2160 RGB, packed uint8 taps from HBM/L2 (stage 3 alone)\n");
    printf("A thread-local (16 taps per lane)            %.3f ms/frame  %8.1f Mpix/s\n", ms[0], mpix / ms[0] * 1e3);
    printf("B cross-lane (16 lanes per output, DPP sums) %.3f ms/frame  %8.1f Mpix/s   (%.2fx the time of A)\n", ms[1], mpix / ms[1] * 1e3, ms[1] / ms[0]);
    printf("outputs: %zu of %zu bytes differ (summation order), max |diff| %d\n", diff, ha.size(), maxd);
    return 0;
}
