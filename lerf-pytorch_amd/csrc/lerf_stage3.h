// Stage-3 (spatially-varying resampling) float32 arithmetic shared by the direct
// and the tile-fused kernels, written with explicit operations (no compiler FMA
// contraction) so that both paths produce bit-identical results.
//
// Reference: SteeringGaussianResize2dNumpy.{sk_weight,resize}
// (resize_right/resize_right2d_numpy.py:150-223) and
// AmplifiedLinearResize2dNumpy.{linear_alpha,linear_weight,resize} (:233-282).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lerf {
namespace s3 {

// float32(k)/255 correctly rounded (== numpy's float32 division, eval_lut_sr.py:623-628)
// for integer k in [0,255], without a divide: one Newton correction of k*(1/255).
__device__ __forceinline__ float u8_over_255(float k) {
#pragma clang fp contract(off)
    const float r = 1.0f / 255.0f;
    float q = k * r;
    float rem = __builtin_fmaf(-q, 255.0f, k);
    return __builtin_fmaf(rem, r, q);
}

// quadratic form of the steering Gaussian; parameters formed in float32 (:168-170)
__device__ __forceinline__ float gauss_form(float h0, float h1, float h2, float max_sigma, float dx, float dy) {
#pragma clang fp contract(off)
    const float rho = h0 * 2.0f - 1.0f;
    const float sx = h1 * max_sigma, sy = h2 * max_sigma;
    const float tx = sx * dx, ty = sy * dy;
    // (sx dx)^2 - 2 rho (sx dx sy dy) + (sy dy)^2            (:154-158)
    float e = tx * tx;
    e = __builtin_fmaf(-2.0f * rho, tx * ty, e);
    return __builtin_fmaf(ty, ty, e);
}

// Same form from uint8 hyper numerators k0,k1,k2 (h = k/255).  The reference's float32 chain
// fl(2*fl(k/255)-1), fl(max_sigma*fl(k/255)) is followed to within one ulp with one operation per
// parameter; the float32 products below carry the same relative error, so nothing is lost.
// ms255 = max_sigma * (1/255).
__device__ __forceinline__ float gauss_form_u8(float k0, float k1, float k2, float ms255, float dx, float dy) {
#pragma clang fp contract(off)
    const float rho = __builtin_fmaf(k0, 2.0f / 255.0f, -1.0f);
    const float tx = (k1 * ms255) * dx, ty = (k2 * ms255) * dy;
    float e = tx * tx;
    e = __builtin_fmaf(-2.0f * rho, tx * ty, e);
    return __builtin_fmaf(ty, ty, e);
}

__device__ __forceinline__ float lin_alpha_u8(float k0, float ms255) {
#pragma clang fp contract(off)
    return __builtin_fmaf(k0, 2.0f * ms255, -ms255 * 255.0f);
}

// exp(-(e - emin)/2): the largest weight of a support is exactly 1
__device__ __forceinline__ float gauss_weight(float e, float emin) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_exp2f((emin - e) * 0.72134752044448170368f);   // 0.5 * log2(e)
}

__device__ __forceinline__ float lin_alpha_of(float h0, float max_sigma) {
#pragma clang fp contract(off)
    return max_sigma * (h0 * 2.0f - 1.0f);                                 // :249-250
}

// cls: 1 for x in [-1,0), 2 for x in [0,1], 0 otherwise (decided on the float64 distance)
__device__ __forceinline__ int dist_class_f(float x) {
    return (x >= -1.0f && x < 0.0f) ? 1 : ((x >= 0.0f && x <= 1.0f) ? 2 : 0);
}

__device__ __forceinline__ float lin_factor(float alpha, float x, int cls) {
#pragma clang fp contract(off)
    float f = cls == 1 ? alpha * x + 1.0f : (cls == 2 ? 1.0f - alpha * x : 0.0f);
    return f < 0.0f ? 0.0f : f;                                            // negative weights clipped (:240)
}

// normalised weighted sum over N taps; e[] are quadratic forms (GAUSS) or weights
template <bool GAUSS, int N, bool FAST = false>
__device__ __forceinline__ float finish(const float (&e)[N], const float (&v)[N]) {
#pragma clang fp contract(off)
    float num = 0.0f, den = 0.0f;
    if (GAUSS) {
        float emin = e[0];
#pragma unroll
        for (int k = 1; k < N; ++k) emin = fminf(emin, e[k]);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float w = gauss_weight(e[k], emin);
            num = __builtin_fmaf(w, v[k], num);
            den += w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            num = __builtin_fmaf(e[k], v[k], num);
            den += e[k];
        }
    }
    if (FAST) {
        // uint8 outputs: reciprocal + one Newton step instead of the IEEE division sequence
        float r = __builtin_amdgcn_rcpf(den);
        r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
        return num * r;
    }
    return num / den;
}

// clip(np.round(v), 0, 255).astype(uint8)  (eval_lut_sr.py:663-665); NaN -> 0
__device__ __forceinline__ uint8_t to_u8(float v) {
    float r = __builtin_rintf(v);
    r = fminf(fmaxf(r, 0.0f), 255.0f);       // fmaxf(NaN, 0) = 0
    return (uint8_t)(int)r;
}

}  // namespace s3
}  // namespace lerf
