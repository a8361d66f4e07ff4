// Fine-tuning path: the float32 LUT pass of the reference's trainable twin
// SWF2LUT.InterpTorchBatch (resample/model.py:172-385) and its backward.
//
// Forward: LUTq = clamp(rne(127 * weight), -127, 127); out = sum_n w_n * LUTq[idx_n] / q, the 4-simplex walk taking
// its MSBs from the mode's pattern pixels and its LSBs from `lsb` pattern pixels -- for modes c and t the reference
// reads the LSBs at the 'y' pattern positions (:229-232, :240-243); that is reproduced, not repaired.  All products and
// sums are integers below 2^24, so float32 is exact and the result equals the reference bit for bit.
//
// Backward (what torch autograd derives for the reference code):
//   d out / d weight[idx_n][oc] = 127 * w_n / q   where -127 <= rne(127 weight) <= 127 (round = BPDA identity, clamp gate)
//   d out / d img[LSB pixel of sorted axis n] = (P_{n+1} - P_n) / q      (torch.remainder passes the gradient through,
//   floor_divide does not).  Ties between LSBs: the reference's case chain puts the LATER axis first (fab = fa > fb ...),
//   so the sort key carries the axis number below the LSB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lerf_common.h"

namespace lerf {
namespace train {

struct Pattern { int dy[4], dx[4], ldy[4], ldx[4]; };

static bool make_pattern(char mode, Pattern* p) {
    int8_t dy[4], dx[4], ly[4], lx[4];
    if (!mode_pattern(mode, dy, dx)) return false;
    // LSB source pixels: own pattern for s, d, y; the 'y' pattern for c and t (model.py:229-232, 240-243)
    mode_pattern((mode == 'c' || mode == 't') ? 'y' : mode, ly, lx);
    for (int k = 0; k < 4; ++k) { p->dy[k] = dy[k]; p->dx[k] = dx[k]; p->ldy[k] = ly[k]; p->ldx[k] = lx[k]; }
    return true;
}

__device__ __forceinline__ float lutq(float w) { return fminf(fmaxf(rintf(w * 127.0f), -127.0f), 127.0f); }

struct Walk {
    int idx[5];     // LUT rows of the 5 corners
    int w[5];       // integer weights, sum = q
    int axis[4];    // axis (0..3 = a..d) stepped at sorted position n
};

// keys: (LSB << 4 | axis) << 16 | stride, sorted descending -> later axis first among equal LSBs
__device__ __forceinline__ Walk walk(const int m[4], const int f[4]) {
    unsigned k0 = ((unsigned)(f[0] << 4 | 0) << 16) | (unsigned)kStrideA;
    unsigned k1 = ((unsigned)(f[1] << 4 | 1) << 16) | (unsigned)kStrideB;
    unsigned k2 = ((unsigned)(f[2] << 4 | 2) << 16) | (unsigned)kStrideC;
    unsigned k3 = ((unsigned)(f[3] << 4 | 3) << 16) | (unsigned)kStrideD;
    ce_desc(k0, k1);
    ce_desc(k2, k3);
    ce_desc(k0, k2);
    ce_desc(k1, k3);
    ce_desc(k1, k2);
    const unsigned k[4] = {k0, k1, k2, k3};
    Walk r;
    r.idx[0] = m[0] * kStrideA + m[1] * kStrideB + m[2] * kStrideC + m[3];
    int fs[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        r.idx[n + 1] = r.idx[n] + (int)(k[n] & 0xffffu);
        fs[n] = (int)(k[n] >> 20);
        r.axis[n] = (int)((k[n] >> 16) & 3u);
    }
    r.w[0] = kQ - fs[0];
    r.w[1] = fs[0] - fs[1];
    r.w[2] = fs[1] - fs[2];
    r.w[3] = fs[2] - fs[3];
    r.w[4] = fs[3];
    return r;
}

__device__ __forceinline__ Walk walk_at(const float* __restrict__ plane, int wp, int y, int x, const Pattern& pt) {
    int m[4], f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int vm = (int)plane[(y + pt.dy[k]) * wp + x + pt.dx[k]];
        const int vl = (int)plane[(y + pt.ldy[k]) * wp + x + pt.ldx[k]];
        m[k] = vm >> 4;          // floor_divide(img, q)
        f[k] = vl & 15;          // img % q
    }
    return walk(m, f);
}

template <int OC>
__global__ void __launch_bounds__(256)
interp_fwd_kernel(const float* __restrict__ weight, const float* __restrict__ img, int n_planes, int h, int w, int bd,
                  Pattern pt, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, p = blockIdx.z;
    if (x >= w) return;
    const int wp = w + bd;
    const Walk r = walk_at(img + (int64_t)p * (h + bd) * wp, wp, y, x, pt);
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) {
        float acc = 0.0f;
#pragma unroll
        for (int n = 0; n < 5; ++n) acc += (float)r.w[n] * lutq(weight[(int64_t)r.idx[n] * OC + oc]);
        out[(((int64_t)p * OC + oc) * h + y) * w + x] = acc / (float)kQ;
    }
}

template <int OC>
__global__ void __launch_bounds__(256)
interp_bwd_kernel(const float* __restrict__ weight, const float* __restrict__ img, const float* __restrict__ gout, int n_planes,
                  int h, int w, int bd, Pattern pt, float* __restrict__ gweight, float* __restrict__ gimg) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, p = blockIdx.z;
    if (x >= w) return;
    const int wp = w + bd;
    const int64_t poff = (int64_t)p * (h + bd) * wp;
    const Walk r = walk_at(img + poff, wp, y, x, pt);
    float gf[4] = {0.0f, 0.0f, 0.0f, 0.0f};       // d loss / d f at sorted position n
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) {
        const float g = gout[(((int64_t)p * OC + oc) * h + y) * w + x] / (float)kQ;
        float P[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const float wv = weight[(int64_t)r.idx[n] * OC + oc];
            const float rq = rintf(wv * 127.0f);
            P[n] = fminf(fmaxf(rq, -127.0f), 127.0f);
            if (gweight && r.w[n] != 0 && rq >= -127.0f && rq <= 127.0f)
                atomicAdd(gweight + (int64_t)r.idx[n] * OC + oc, g * (float)r.w[n] * 127.0f);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) gf[n] += g * (P[n + 1] - P[n]);
    }
    if (gimg) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int a = r.axis[n];
            atomicAdd(gimg + poff + (y + pt.ldy[a]) * wp + x + pt.ldx[a], gf[n]);
        }
    }
}

}  // namespace train
}  // namespace lerf

using namespace lerf::train;

extern "C" {

int lerf_swf2lut_interp_f32(const float* weight, int oC, char mode, const float* img, int n_planes, int h, int w, int bd,
                            float* out, void* stream) {
    Pattern pt;
    if (!weight || !img || !out || n_planes < 1 || h < 1 || w < 1 || bd < 0) return LERF_EINVAL;
    if (!make_pattern(mode, &pt)) return LERF_EINVAL;
    for (int k = 0; k < 4; ++k)
        if (pt.dy[k] > bd || pt.dx[k] > bd || pt.ldy[k] > bd || pt.ldx[k] > bd) return LERF_EINVAL;
    dim3 block(256), grid((w + 255) / 256, h, n_planes);
    hipStream_t st = (hipStream_t)stream;
    if (oC == 1) hipLaunchKernelGGL(interp_fwd_kernel<1>, grid, block, 0, st, weight, img, n_planes, h, w, bd, pt, out);
    else if (oC == 3) hipLaunchKernelGGL(interp_fwd_kernel<3>, grid, block, 0, st, weight, img, n_planes, h, w, bd, pt, out);
    else return LERF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? LERF_OK : LERF_ELAUNCH;
}

int lerf_swf2lut_interp_bwd_f32(const float* weight, int oC, char mode, const float* img, const float* grad_out, int n_planes,
                                int h, int w, int bd, float* grad_weight, float* grad_img, void* stream) {
    Pattern pt;
    if (!weight || !img || !grad_out || n_planes < 1 || h < 1 || w < 1 || bd < 0) return LERF_EINVAL;
    if (!make_pattern(mode, &pt)) return LERF_EINVAL;
    for (int k = 0; k < 4; ++k)
        if (pt.dy[k] > bd || pt.dx[k] > bd || pt.ldy[k] > bd || pt.ldx[k] > bd) return LERF_EINVAL;
    dim3 block(256), grid((w + 255) / 256, h, n_planes);
    hipStream_t st = (hipStream_t)stream;
    if (oC == 1)
        hipLaunchKernelGGL(interp_bwd_kernel<1>, grid, block, 0, st, weight, img, grad_out, n_planes, h, w, bd, pt, grad_weight, grad_img);
    else if (oC == 3)
        hipLaunchKernelGGL(interp_bwd_kernel<3>, grid, block, 0, st, weight, img, grad_out, n_planes, h, w, bd, pt, grad_weight, grad_img);
    else
        return LERF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? LERF_OK : LERF_ELAUNCH;
}

}  // extern "C"
