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

// uint8 production arithmetic (uint8 in -> uint8 out, both the direct and the fused kernel):
// the quadratic form is evaluated pre-scaled by 0.5*log2(e) so that weight = exp2(emin' - e').
//   dxs = dx * (max_sigma/255) * sqrt(0.5*log2 e)   (per row tap; gauss_scale())
//   dys = dy * (max_sigma/255) * sqrt(0.5*log2 e)   (per column tap)
//   e'  = (k1 dxs)^2 + (k2 dys)^2 - 2 rho (k1 dxs)(k2 dys),  -2 rho = 2 - 4 k0/255
// The reference forms rho, sigma in float32 from h = fl(k/255) (resize_right2d_numpy.py:168-170);
// one fused operation per parameter follows that chain to within an ulp, the same size as the
// rounding of the float32 products that consume it.
__device__ __forceinline__ float gauss_scale(float max_sigma) {
#pragma clang fp contract(off)
    return (max_sigma * (1.0f / 255.0f)) * 0.84932180028801904272f;        // sqrt(0.5 * log2(e))
}

// the pieces of gauss_form_u8, so that callers can hoist what several outputs share (bit-identical results)
__device__ __forceinline__ float gauss_m2rho_u8(float k0) {
#pragma clang fp contract(off)
    return __builtin_fmaf(k0, -4.0f / 255.0f, 2.0f);
}
__device__ __forceinline__ float gauss_t_u8(float k, float ds) {
#pragma clang fp contract(off)
    return k * ds;
}
// e' = tx^2 + ty^2 - 2 rho tx ty with the two column-only terms ty^2 and (-2 rho ty) formed first, so that output rows
// sharing their taps pay three operations per tap (tx, two FMAs)
__device__ __forceinline__ float gauss_form_cols(float tx, float ty2, float mty) {
#pragma clang fp contract(off)
    return __builtin_fmaf(tx, mty, __builtin_fmaf(tx, tx, ty2));
}
__device__ __forceinline__ float gauss_form_parts(float m2rho, float tx, float ty) {
#pragma clang fp contract(off)
    return gauss_form_cols(tx, ty * ty, m2rho * ty);
}
__device__ __forceinline__ float gauss_form_u8(float k0, float k1, float k2, float dxs, float dys) {
    return gauss_form_parts(gauss_m2rho_u8(k0), gauss_t_u8(k1, dxs), gauss_t_u8(k2, dys));
}

// exp2(emin' - e') for pre-scaled forms
__device__ __forceinline__ float gauss_weight_scaled(float e, float emin) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_exp2f(emin - e);
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

// the same factor from |x| and its inside-the-support mask (1 for |x| <= 1 -- both classes of dist_class_f --, 0 outside):
// alpha * x + 1 for x < 0 and 1 - alpha * x for x >= 0 are both fl(1 - fl(alpha |x|)), so the result is lin_factor's, bit
// for bit; callers that meet a distance many times (one row or column of a block of outputs) keep |x| and the mask
__device__ __forceinline__ float lin_factor_abs(float alpha, float ax, float mask) {
#pragma clang fp contract(off)
    const float f = 1.0f - alpha * ax;
    return (f < 0.0f ? 0.0f : f) * mask;
}

// The float32 production arithmetic of the uint8 paths (pre-scaled quadratic forms, unshifted exp2 sums, tie guard of
// 1.5e-4) is sized for max_sigma <= kNoShiftMaxSigma: on an SR grid the nearest tap lies within half a pixel on both
// axes, so its pre-scaled form is at most 0.5 log2(e) (0.25 + 0.25 + 0.5) max_sigma^2 = 0.7214 max_sigma^2 < 126 and
// exp2 of it stays a normal float32 (the denominator cannot vanish), and the float32 rounding of forms of that size stays
// far inside the tie guard.  max_sigma is a free parameter of the reference's classes (resize_right2d_numpy.py:143,
// option.py:29 default 10) and of the C ABI: the HOST sends larger values through the float64 direct kernel
// (launch_resize, fused_supported), whose arithmetic is the reference's own.
constexpr float kNoShiftMaxSigma = 13.0f;

// num / den of the uint8 paths: reciprocal + one Newton step instead of the IEEE division sequence (finish<.., FAST>)
__device__ __forceinline__ float finish_div(float num, float den) {
#pragma clang fp contract(off)
    float r = __builtin_amdgcn_rcpf(den);
    r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
    return num * r;
}

// normalised weighted sum over N taps; e[] are quadratic forms (GAUSS) or weights
template <bool GAUSS, int N, bool FAST = false, bool SCALED = false, bool NOSHIFT = false>
__device__ __forceinline__ float finish(const float (&e)[N], const float (&v)[N]) {
#pragma clang fp contract(off)
    // the sums start from the first tap (0 + x and fma(w, v, 0) are exact, but without fast-math the compiler keeps them)
    float num, den;
    if (GAUSS && SCALED && NOSHIFT && (N == 4 || N == 16)) {
        // 2x2 / 4x4 support on an SR grid (taps at left .. left+S-1 with left = ceil(g - S/2 - eps)), max_sigma <=
        // kNoShiftMaxSigma: the nearest tap's pre-scaled form is at most 0.7214 * 169 = 122 (72 at the default max_sigma
        // of 10) and exp2(-122) is a normal float32 -- the denominator cannot vanish and the weights need no shift by the support's
        // minimum (N-1 v_min + N v_sub per output saved; the relative accuracy of w = exp2(-e) is that of e either way)
        den = __builtin_amdgcn_exp2f(-e[0]);
        num = den * v[0];
#pragma unroll
        for (int k = 1; k < N; ++k) {
            const float w = __builtin_amdgcn_exp2f(-e[k]);
            num = __builtin_fmaf(w, v[k], num);
            den += w;
        }
    } else if (GAUSS) {
        float emin = e[0];
#pragma unroll
        for (int k = 1; k < N; ++k) emin = fminf(emin, e[k]);
        den = SCALED ? gauss_weight_scaled(e[0], emin) : gauss_weight(e[0], emin);
        num = den * v[0];
#pragma unroll
        for (int k = 1; k < N; ++k) {
            const float w = SCALED ? gauss_weight_scaled(e[k], emin) : gauss_weight(e[k], emin);
            num = __builtin_fmaf(w, v[k], num);
            den += w;
        }
    } else {
        den = e[0];
        num = e[0] * v[0];
#pragma unroll
        for (int k = 1; k < N; ++k) {
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

// ---------------------------------------------------------------------------
// Packed float32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per instruction, the same
// results lane for lane as the scalar forms).  In the kernels' mixed instruction streams every VALU instruction costs one
// 4-cycle pass (profiles/r03_issue_rates.txt), a packed one ~1.4 passes for two operations.  A scalar that both lanes use is
// read through op_sel from the LOW or HIGH half of a register pair -- no copy into a pair of its own.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
// (a.x * b.x, a.y * b.x)
__device__ __forceinline__ f2 pk_mul_blo(f2 a, f2 b) {
    f2 d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (a.L * a.L + c.x, a.L * a.L + c.y) and (a.L * b.x + c.x, a.L * b.y + c.y) with L = the low (HI = false) or high half of a
template <bool HI>
__device__ __forceinline__ f2 pk_fma_aa(f2 a, f2 c) {
    f2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
template <bool HI>
__device__ __forceinline__ f2 pk_fma_ab(f2 a, f2 b, f2 c) {
    f2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// (a.x * b.x + c.x, a.y * b.x + c.y)
__device__ __forceinline__ f2 pk_fma_blo(f2 a, f2 b, f2 c) {
    f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// pairs a (two lanes) against the scalars packed in b: (a.x * a.x + b.H, a.y * a.y + b.H) and (a.x * b.H + c.x, a.y * b.H + c.y)
template <bool HI>
__device__ __forceinline__ f2 pk_fma_ppb(f2 a, f2 b) {
    f2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <bool HI>
__device__ __forceinline__ f2 pk_fma_pb(f2 a, f2 b, f2 c) {
    f2 d;
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the same with b's HIGH half, for operands that a transcendental instruction has just written: gfx950 needs one wait state
// between v_exp / v_rcp and a VALU read of the result, which the compiler inserts for its own instructions only -- these
// carry it themselves (s_nop 0)
__device__ __forceinline__ f2 pk_mul_bhi_after_trans(f2 w, f2 b) {
    f2 d;
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(w), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pk_fma_bhi_after_trans(f2 w, f2 b, f2 c) {
    f2 d;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) {
    f2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) {
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ---------------------------------------------------------------------------
// Tie guard for uint8 outputs.  The float32 result can differ from the reference's float64 result
// by up to ~1e-4 (0..255 scale), which flips clip(round(.)) only when the value sits that close to a
// half-integer.  Such values (about 1 in 2500) are re-evaluated in float64 with the reference's own
// dtype chain -- float32 parameter formation, everything else float64, same tap order
// (resize_right2d_numpy.py:150-160, 168-170, 200-221) -- so the uint8 output equals the reference's
// byte for byte unless the float32 error exceeds kTieEps = 1.5e-4 (the float32 path is typically within 5e-6, 9e-5 at worst in the tests).
// ---------------------------------------------------------------------------
#ifndef LERF_TIE_EPS                                                  // -DLERF_TIE_EPS=1e-3f: the strict build (see below); A/B in profiles/r06_tie_eps_ab.txt
#define LERF_TIE_EPS 1.5e-4f
#endif
// DETECTION threshold: outputs this close to a half-integer leave the float32 path (resolve_u8 below).  1.5e-4 covers the
// float32 error of every map the LUTs produce on the test sets (0 of 6.9 G bytes); forms of ~70 (sigma saturated over a whole
// support) carry float32 rounding of 2e-4 .. 1e-3 of the 0..255 scale: a build with 1e-3 is byte-exact on those too (all but
// exact float64 ties, where one ulp of exp() decides) at 0.7 % of the SR throughput, 2.7 % of the warp's (seven times as many
// outputs detected).
constexpr float kTieEps = LERF_TIE_EPS;
// the part of them that needs the reference's own float64 exp(): closer than this after the intermediate evaluation
constexpr float kTieEpsExact = 1.5e-4f;

__device__ __forceinline__ bool near_tie(float x) {
    return __builtin_fabsf(x - __builtin_rintf(x)) > 0.5f - kTieEps;
}

// byte BYTE of `packed` <- clip(rne(x), 0, 255) with one v_cvt_pk_u8_f32 (saturating convert + byte insert), and
// the tie test on the same v_rndne
__device__ __forceinline__ uint32_t pack_u8_tie(float x, int byte, uint32_t packed, bool* tie) {
    const float r = __builtin_rintf(x);
    *tie = __builtin_fabsf(x - r) > 0.5f - kTieEps;
    return __builtin_amdgcn_cvt_pk_u8_f32(r, (unsigned)byte, packed);
}
// the same, handing back the distance from the rounded value: callers with many outputs keep a running maximum of
// |dist| (one v_max per output) and look for the ties (|dist| > 0.5 - kTieEps) only when that maximum says there is one
__device__ __forceinline__ uint32_t pack_u8_dist(float x, int byte, uint32_t packed, float* dist) {
    const float r = __builtin_rintf(x);
    *dist = x - r;
    return __builtin_amdgcn_cvt_pk_u8_f32(r, (unsigned)byte, packed);
}

// to_u8 and the tie test sharing one v_rndne
__device__ __forceinline__ uint32_t to_u8_tie(float x, bool* tie) {
    const float r = __builtin_rintf(x);
    *tie = __builtin_fabsf(x - r) > 0.5f - kTieEps;
    return (uint32_t)(int)fminf(fmaxf(r, 0.0f), 255.0f);
}

// d[a*S+b] = (k0 | k1<<8 | k2<<16 | val<<24) of tap (column offset a, row offset b); dx[b], dy[a] float64
template <bool GAUSS, int S>
__device__ __forceinline__ double eval64(const uint32_t (&d)[S * S], const double (&dx)[S], const double (&dy)[S],
                                         float max_sigma) {
#pragma clang fp contract(off)
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const uint32_t q = d[a * S + b];
            const float h0 = u8_over_255((float)(q & 0xFFu));
            double w;
            if (GAUSS) {
                const float h1 = u8_over_255((float)((q >> 8) & 0xFFu));
                const float h2 = u8_over_255((float)((q >> 16) & 0xFFu));
                const double rho = (double)(h0 * 2.0f - 1.0f);
                const double sx = (double)(h1 * max_sigma), sy = (double)(h2 * max_sigma);
                const double xn = (sx * dx[b]) * (sx * dx[b]);
                const double yn = (sy * dy[a]) * (sy * dy[a]);
                const double xy = sx * dx[b] * sy * dy[a];
                w = exp(-0.5 * (xn - 2.0 * rho * xy + yn));
            } else {
                const double al = (double)(max_sigma * (h0 * 2.0f - 1.0f));
                const double x = dx[b], y = dy[a];
                double fx = 0.0, fy = 0.0;
                if (x >= -1.0 && x < 0.0) fx = al * x + 1.0; else if (x >= 0.0 && x <= 1.0) fx = 1.0 - al * x;
                if (y >= -1.0 && y < 0.0) fy = al * y + 1.0; else if (y >= 0.0 && y <= 1.0) fy = 1.0 - al * y;
                w = (fx < 0.0 ? 0.0 : fx) * (fy < 0.0 ? 0.0 : fy);
            }
            num += w * (double)(q >> 24);
            den += w;
        }
    return num / den;
}

__device__ __forceinline__ uint8_t to_u8_d(double v) {
    double r = __builtin_rint(v);
    r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);     // NaN compares false twice -> stays NaN -> (int) below gives 0 on gfx950
    return (v != v) ? (uint8_t)0 : (uint8_t)(int)r;
}

// Two-level resolution of an output the float32 path could not decide (|x - rint(x)| > 0.5 - kTieEps).
// Level 1 (Gaussian kind): the reference's float64 quadratic forms, bit for bit those of eval64, and weights from the hardware
// exp2 of their float64 DIFFERENCES to the smallest form -- what the float32 path loses is the rounding of forms of size ~70
// (4 ulp of the form, up to 1e-3 of the output); a difference that matters (weight above 1e-6) is below 20, its float32
// rounding 1e-6, the weight's error 1e-6 relative, the output's below 1e-5.  Level 1 decides unless its own result is within
// kTieEpsExact of a half-integer; then, as before round 6, the reference's own dtype chain with the float64 exp() (eval64).
// The linear kind has no exponent to lose: eval64 directly (no exp in it).
template <bool GAUSS, int S>
__device__ __forceinline__ uint8_t resolve_u8(const uint32_t (&d)[S * S], const double (&dx)[S], const double (&dy)[S],
                                              float max_sigma) {
#pragma clang fp contract(off)
    // (S = 4: sixteen forms twice over cost the kernels registers they do not have; detection at the exact threshold -- the shipped
    //  build -- leaves level 1 nothing to decide: it would only delay the float64 chain)
    if constexpr (GAUSS && S == 2 && (kTieEps > kTieEpsExact)) {
        auto form = [&](int a, int b) -> double {                       // eval64's argument of exp(-0.5 .), same operations
            const uint32_t q = d[a * S + b];
            const float h0 = u8_over_255((float)(q & 0xFFu));
            const float h1 = u8_over_255((float)((q >> 8) & 0xFFu));
            const float h2 = u8_over_255((float)((q >> 16) & 0xFFu));
            const double rho = (double)(h0 * 2.0f - 1.0f);
            const double sx = (double)(h1 * max_sigma), sy = (double)(h2 * max_sigma);
            const double xn = (sx * dx[b]) * (sx * dx[b]);
            const double yn = (sy * dy[a]) * (sy * dy[a]);
            const double xy = sx * dx[b] * sy * dy[a];
            return xn - 2.0 * rho * xy + yn;
        };
        double emin = form(0, 0);
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) emin = __builtin_fmin(emin, form(a, b));
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) {
                const float t = (float)((emin - form(a, b)) * 0.72134752044448170368);   // 0.5 log2(e): <= 0, 0 for the smallest form
                const double w = (double)__builtin_amdgcn_exp2f(t);
                num += w * (double)(d[a * S + b] >> 24);
                den += w;                                                 // >= 1
            }
        const double mid = num / den;
        const double r = __builtin_rint(mid);
        if (__builtin_fabs(mid - r) <= 0.5 - (double)kTieEpsExact)
            return (uint8_t)(int)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
    }
    return to_u8_d(eval64<GAUSS, S>(d, dx, dy, max_sigma));
}

// clip(np.round(v), 0, 255).astype(uint8)  (eval_lut_sr.py:663-665); NaN -> 0
__device__ __forceinline__ uint8_t to_u8(float v) {
    float r = __builtin_rintf(v);
    r = fminf(fmaxf(r, 0.0f), 255.0f);       // fmaxf(NaN, 0) = 0
    return (uint8_t)(int)r;
}

}  // namespace s3
}  // namespace lerf
