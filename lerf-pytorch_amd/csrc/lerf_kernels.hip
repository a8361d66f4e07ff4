// LeRF LUT resampling kernels for MI355X (gfx950, wave64).  Hand-written HIP;
// no CUDA compatibility paths.  The C ABI (include/lerf_hip.h) is implemented
// at the bottom of lerf_api.hip; this file holds the general-purpose kernels:
//
//   (A1, one (LUT, pattern) simplex pass: lerf_lut_interp.hip)
//   lut_stage_kernel    rotation/mode ensemble + rounding      A2/A3 (eval_lut_sr.py:541-628)
//   resize_kernel       separable-geometry stage 3 (SR)        A5/A6 (resize_right2d_numpy.py:142-282)
//   warp_kernel         homography stage 3                     A7/A8 (resize_right2d_numpy.py:284-636)
//
// These "direct" kernels gather the LUTs from global memory (L1/L2) and keep
// every tensor in HBM between stages; they serve the class-level API (float
// CHW tensors, arbitrary modes / support sizes).  The tile-fused uint8 path
// lives in lerf_fused.hip.
#include "lerf_kernels.h"
#include "lerf_stage3.h"
#include "lerf_warp_px.h"

namespace lerf {

// ---------------------------------------------------------------------------
// A2/A3: one LUT stage (sum over modes x 4 rotations, divide, bias, round, clip)
// ---------------------------------------------------------------------------
template <int OC>
__global__ void __launch_bounds__(256)
lut_stage_kernel(const uint8_t* __restrict__ img, int64_t sy, int64_t sx, int64_t sc,
                 int H, int W, int C, StageLuts luts, int div, int bias,
                 uint8_t* __restrict__ out, int64_t oy, int64_t ox, int64_t oc_stride) {
    // x runs over (column, channel) so that HWC frames are read/written densely
    int xc = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (xc >= W * C) return;
    int x = xc / C;
    int c = xc - x * C;
    const uint8_t* plane = img + c * sc;
    int acc[OC];
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) acc[oc] = 0;
    for (int m = 0; m < luts.n_modes; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int8_t* __restrict__ lut = luts.lut[m][r & 1];
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int yy = clampi(y + luts.off[m][r].dy[k], 0, H - 1);
                int xx = clampi(x + luts.off[m][r].dx[k], 0, W - 1);
                v[k] = plane[yy * sy + xx * sx];
            }
            SimplexPath p = simplex_path(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
#pragma unroll
                for (int oc = 0; oc < OC; ++oc) acc[oc] += p.w[n] * (int)lut[p.idx[n] * OC + oc];
            }
        }
    }
    uint8_t* o = out + y * oy + x * ox + c * oc_stride;
#pragma unroll
    for (int oc = 0; oc < OC; ++oc) o[oc] = (uint8_t)rne_div_clip255(acc[oc] + bias * div, div);
}

int launch_lut_stage(const uint8_t* img, int64_t sy, int64_t sx, int64_t sc, int H, int W, int C,
                     const StageLuts& luts, int oC, int div, int bias,
                     uint8_t* out, int64_t oy, int64_t ox, int64_t ocs, hipStream_t st) {
    dim3 block(256), grid((W * C + 255) / 256, H);
    if (oC == 1)
        hipLaunchKernelGGL(lut_stage_kernel<1>, grid, block, 0, st, img, sy, sx, sc, H, W, C, luts, div, bias, out, oy, ox, ocs);
    else if (oC == 3)
        hipLaunchKernelGGL(lut_stage_kernel<3>, grid, block, 0, st, img, sy, sx, sc, H, W, C, luts, div, bias, out, oy, ox, ocs);
    else
        return LERF_EUNSUPPORTED;
    return LERF_OK;
}

// ---------------------------------------------------------------------------
// stage 3 building blocks
// ---------------------------------------------------------------------------
template <typename T> struct Loader;
template <> struct Loader<uint8_t> {
    // hyper numerators: h = float32(u8) / 255 exactly as eval_lut_sr.py:623-628
    static __device__ __forceinline__ float hyper(const uint8_t* p) { return s3::u8_over_255((float)(*p)); }
    static __device__ __forceinline__ float pixel(const uint8_t* p) { return (float)(*p); }
};
template <> struct Loader<float> {
    static __device__ __forceinline__ float hyper(const float* p) { return *p; }
    static __device__ __forceinline__ float pixel(const float* p) { return *p; }
};

template <typename T> struct Storer;
template <> struct Storer<uint8_t> {
    // clip(np.round(x), 0, 255).astype(uint8)  (eval_lut_sr.py:663-665); NaN -> 0
    template <typename A> static __device__ __forceinline__ void put(uint8_t* p, A v) {
        if (sizeof(A) == 8) *p = s3::to_u8_d((double)v);      // float64 arithmetic: rounded once, from the double
        else *p = s3::to_u8((float)v);
    }
};
template <> struct Storer<float> {
    template <typename A> static __device__ __forceinline__ void put(float* p, A v) { *p = (float)v; }
};
template <> struct Storer<double> {
    template <typename A> static __device__ __forceinline__ void put(double* p, A v) { *p = (double)v; }
};

// Tap accumulator.  A = float: exponents are shifted by their minimum over the
// support before exp (result unchanged mathematically; the largest weight is 1
// so the normalisation can never be 0/0 where the float64 reference is finite).
template <typename A, int KIND, int MAXT>
struct TapAcc {
    A e[MAXT];   // gauss: quadratic form; linear/nearest: weight
    A v[MAXT];
    int n = 0;

    __device__ __forceinline__ void add_gauss(A rho, A sx, A sy, A dx, A dy, A val) {
#pragma clang fp contract(off)
        // resize_right2d_numpy.py:150-160, in the reference's operation order and without FMA contraction: the uint8 outputs
        // of the float64 path (max_sigma > 13) must round like numpy's even where two taps carry EQUAL weights and the
        // value is an exact half (x_norm = (sx dx)^2, y_norm = (sy dy)^2, xy = sx dx sy dy left to right)
        const A xn = (sx * dx) * (sx * dx), yn = (sy * dy) * (sy * dy), xy = sx * dx * sy * dy;
        e[n] = xn - (A)2 * rho * xy + yn;
        v[n] = val;
        ++n;
    }
    __device__ __forceinline__ void add_weight(A w, A val) {
        e[n] = w;
        v[n] = val;
        ++n;
    }
    __device__ __forceinline__ A finish() const {
#pragma clang fp contract(off)
        A num = 0, den = 0;
        if (KIND == LERF_KIND_GAUSS) {
            // float32: shifted by the minimum (see above).  float64: NOT shifted -- the reference's own arithmetic
            // (np.exp(-0.5 e), resize_right2d_numpy.py:150-160), including where its weights run into the denormal range
            // for large max_sigma and lose mantissa bits: a shift would be more accurate than the reference, i.e. different
            A emin = e[0];
#pragma unroll
            for (int k = 1; k < MAXT; ++k)
                if (k < n) emin = e[k] < emin ? e[k] : emin;
            if (sizeof(A) == 8) emin = (A)0;
#pragma unroll
            for (int k = 0; k < MAXT; ++k)
                if (k < n) {
                    A w;
                    if (sizeof(A) == 4)
                        w = (A)__expf((float)((A)-0.5 * (e[k] - emin)));
                    else
                        w = (A)exp((double)((A)-0.5 * (e[k] - emin)));
                    num += w * v[k];
                    den += w;
                }
            // float64 reference: every weight underflows to 0 -> 0/0 = NaN (float32: emulated; float64: it happens by itself)
            if (sizeof(A) == 4 && emin * (A)0.5 > (A)745.2) return (A)(0.0 / 0.0);
            return num / den;
        } else {
#pragma unroll
            for (int k = 0; k < MAXT; ++k)
                if (k < n) {
                    num += e[k] * v[k];
                    den += e[k];
                }
            return num / den;   // 0/0 = NaN like the reference
        }
    }
};

// ---------------------------------------------------------------------------
// A5/A6: SR with separable geometry tables
// ---------------------------------------------------------------------------
template <typename TI, typename TH, typename TO, typename A, int KIND, int ST>
__global__ void __launch_bounds__(256)
resize_kernel(const TI* __restrict__ feat, int64_t fy, int64_t fx, int64_t fc,
              const TH* __restrict__ h0, const TH* __restrict__ h1, const TH* __restrict__ h2,
              int64_t hy, int64_t hx, int64_t hc,
              int H, int W, int C, int S_rt, int oH, int oW,
              const int* __restrict__ left_r, const A* __restrict__ dis_r,
              const int* __restrict__ left_c, const A* __restrict__ dis_c,
              const double* __restrict__ dis_r64, const double* __restrict__ dis_c64,
              A max_sigma, TO* __restrict__ out, int64_t oy, int64_t ox, int64_t oc, int pad_mode) {
    const int S = ST > 0 ? ST : S_rt;
    constexpr int MAXS = ST > 0 ? ST : LERF_MAX_SUPPORT;
    int xc = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (xc >= oW * C) return;
    int j = xc / C;
    int c = xc - j * C;
    int lr = left_r[i], lc = left_c[j];
    A num = 0, den = 0;
    // the reference sums column-offset major, row-offset minor (numpy meshgrid 'xy', :95-98)
    if (ST > 0 && sizeof(A) == 4) {
        // float32 production path: same arithmetic as the tile-fused kernel (lerf_stage3.h)
        float e[MAXS * MAXS], v[MAXS * MAXS];
        uint32_t dd[MAXS * MAXS];      // (k0,k1,k2,val) per tap for the tie guard (uint8 in / uint8 out only)
#pragma unroll
        for (int a = 0; a < MAXS; ++a) {
#pragma unroll
            for (int b = 0; b < MAXS; ++b) {
                int rr = lr + b, cc = lc + a;
                int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);
                bool zr, zc;                                                                           // image pad rule (:208)
                const int rs = pad_index(rr, H, pad_mode, &zr), cs = pad_index(cc, W, pad_mode, &zc);
                v[a * MAXS + b] = (zr || zc) ? 0.0f : Loader<TI>::pixel(feat + rs * fy + cs * fx + c * fc);
                if (sizeof(TH) == 1 && sizeof(TO) == 1) {
                    const int64_t hh = rcl * hy + ccl * hx + c * hc;
                    dd[a * MAXS + b] = (uint32_t)h0[hh] | ((KIND == LERF_KIND_GAUSS ? (uint32_t)h1[hh] : 0u) << 8) |
                                       ((KIND == LERF_KIND_GAUSS ? (uint32_t)h2[hh] : 0u) << 16) |
                                       ((uint32_t)v[a * MAXS + b] << 24);
                }
                int64_t ho = rcl * hy + ccl * hx + c * hc;                                                    // edge pad (:172-174)
                float dx = (float)dis_r[i * S + b], dy = (float)dis_c[j * S + a];
                // uint8 in / uint8 out: the production arithmetic of the fused kernel (bit-identical results);
                // float outputs keep the exact float32 parameter formation of the reference
                constexpr bool U8H = sizeof(TH) == 1 && sizeof(TO) == 1;
                const float ms255 = (float)max_sigma * (1.0f / 255.0f);
                if (KIND == LERF_KIND_GAUSS) {
                    if (U8H)
                        e[a * MAXS + b] = s3::gauss_form_u8((float)h0[ho], (float)h1[ho], (float)h2[ho],
                                                            dx * s3::gauss_scale((float)max_sigma),
                                                            dy * s3::gauss_scale((float)max_sigma));
                    else
                        e[a * MAXS + b] = s3::gauss_form(Loader<TH>::hyper(h0 + ho), Loader<TH>::hyper(h1 + ho),
                                                         Loader<TH>::hyper(h2 + ho), (float)max_sigma, dx, dy);
                } else {
                    float alpha = U8H ? s3::lin_alpha_u8((float)h0[ho], ms255)
                                      : s3::lin_alpha_of(Loader<TH>::hyper(h0 + ho), (float)max_sigma);
                    e[a * MAXS + b] = s3::lin_factor(alpha, dx, s3::dist_class_f(dx)) * s3::lin_factor(alpha, dy, s3::dist_class_f(dy));
                }
            }
        }
        // unshifted sums: max_sigma <= s3::kNoShiftMaxSigma on this path (launch_resize sends larger values to the float64 one)
        constexpr bool U8P = sizeof(TH) == 1 && sizeof(TO) == 1;
        const float xf = s3::finish<KIND == LERF_KIND_GAUSS, MAXS * MAXS, sizeof(TO) == 1, U8P, U8P>(e, v);
        if (sizeof(TH) == 1 && sizeof(TO) == 1 && dis_r64 != nullptr && s3::near_tie(xf)) {
            double dx64[MAXS], dy64[MAXS];
#pragma unroll
            for (int b = 0; b < MAXS; ++b) dx64[b] = dis_r64[i * S + b];
#pragma unroll
            for (int a = 0; a < MAXS; ++a) dy64[a] = dis_c64[j * S + a];
            Storer<TO>::put(out + i * oy + j * ox + c * oc,
                            (float)s3::resolve_u8<KIND == LERF_KIND_GAUSS, MAXS>(dd, dx64, dy64, (float)max_sigma));
            return;
        }
        Storer<TO>::put(out + i * oy + j * ox + c * oc, xf);
        return;
    }
    if (ST > 0) {
        TapAcc<A, KIND, MAXS * MAXS> acc;
#pragma unroll
        for (int a = 0; a < MAXS; ++a) {
#pragma unroll
            for (int b = 0; b < MAXS; ++b) {
                int rr = lr + b, cc = lc + a;
                int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);
                bool zr, zc;                                                                           // image pad rule (:208)
                const int rs = pad_index(rr, H, pad_mode, &zr), cs = pad_index(cc, W, pad_mode, &zc);
                A val = (zr || zc) ? (A)0 : (A)Loader<TI>::pixel(feat + rs * fy + cs * fx + c * fc);
                int64_t ho = rcl * hy + ccl * hx + c * hc;                                             // edge pad (:172-174)
                A dx = dis_r[i * S + b], dy = dis_c[j * S + a];
                if (KIND == LERF_KIND_GAUSS) {
                    float p0 = Loader<TH>::hyper(h0 + ho), p1 = Loader<TH>::hyper(h1 + ho), p2 = Loader<TH>::hyper(h2 + ho);
                    // float32 parameter formation (:168-170)
                    float rho = p0 * 2.0f - 1.0f, sx = p1 * (float)max_sigma, sy = p2 * (float)max_sigma;
                    acc.add_gauss((A)rho, (A)sx, (A)sy, dx, dy, val);
                } else {
                    float p0 = Loader<TH>::hyper(h0 + ho);
                    float alpha = (float)max_sigma * (p0 * 2.0f - 1.0f);
                    A w = lin_factor<A>((A)alpha, dx, dist_class(dx)) * lin_factor<A>((A)alpha, dy, dist_class(dy));
                    acc.add_weight(w, val);
                }
            }
        }
        Storer<TO>::put(out + i * oy + j * ox + c * oc, acc.finish());
        return;
    }
    // generic support size: two passes (min exponent, then accumulate)
    A emin = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                int rr = lr + b, cc = lc + a;
                int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);
                bool zr, zc;
                const int rs = pad_index(rr, H, pad_mode, &zr), cs = pad_index(cc, W, pad_mode, &zc);
                int64_t ho = rcl * hy + ccl * hx + c * hc;
                A dx = dis_r[i * S + b], dy = dis_c[j * S + a];
                A w;
                if (KIND == LERF_KIND_GAUSS) {
                    float p0 = Loader<TH>::hyper(h0 + ho), p1 = Loader<TH>::hyper(h1 + ho), p2 = Loader<TH>::hyper(h2 + ho);
                    A rho = (A)(p0 * 2.0f - 1.0f), sx = (A)(p1 * (float)max_sigma), sy = (A)(p2 * (float)max_sigma);
                    A tx = sx * dx, ty = sy * dy;
                    A e = tx * tx - (A)2 * rho * (tx * ty) + ty * ty;
                    if (pass == 0) {
                        emin = sizeof(A) == 8 ? (A)0 : ((a == 0 && b == 0) ? e : (e < emin ? e : emin));   // float64: unshifted, as TapAcc
                        continue;
                    }
                    w = sizeof(A) == 4 ? (A)__expf((float)((A)-0.5 * (e - emin))) : (A)exp((double)((A)-0.5 * (e - emin)));
                } else {
                    if (pass == 0) continue;
                    float p0 = Loader<TH>::hyper(h0 + ho);
                    A alpha = (A)((float)max_sigma * (p0 * 2.0f - 1.0f));
                    w = lin_factor<A>(alpha, dx, dist_class(dx)) * lin_factor<A>(alpha, dy, dist_class(dy));
                }
                A val = (zr || zc) ? (A)0 : (A)Loader<TI>::pixel(feat + rs * fy + cs * fx + c * fc);
                num += w * val;
                den += w;
            }
    }
    Storer<TO>::put(out + i * oy + j * ox + c * oc, num / den);
}

// ---------------------------------------------------------------------------
// The uint8 production form of the same resampler by SOURCE CELL (S = 2): the outputs whose first tap is (lr, lc) -- about
// scale_h x scale_w of them -- share their four taps.  One thread per (cell, channel) loads and converts the taps once
// (16 byte loads, the per-tap terms of the quadratic form) and walks its outputs: per output column the column-only terms
// (ty^2, -2 rho ty: s3::gauss_form_cols), per output 4 x (tx, two FMAs, exp2) + the normalisation -- a quarter of the
// per-pixel kernel's instructions at x2 (every thread of resize_kernel repeats the loads, the index arithmetic and the
// conversions of its own four taps).  Same operations in the same order on every output: byte-identical results, same
// float64 tie guard.  The cell's outputs are found in the geometry tables themselves (left_r / left_c are nondecreasing:
// a guess from the scale factor, corrected by walking), so the C ABI and its tables stay as they are.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int first_at_least(const int* __restrict__ t, int n, int l, int guess) {
    int i = min(max(guess, 0), n);
    while (i > 0 && t[i - 1] >= l) --i;
    while (i < n && t[i] < l) ++i;
    return i;
}

template <int KIND>
__global__ void __launch_bounds__(256)
resize_cells_u8_kernel(const uint8_t* __restrict__ feat, int fy, int fx, int fc,
                       const uint8_t* __restrict__ h0, const uint8_t* __restrict__ h1, const uint8_t* __restrict__ h2,
                       int hy, int hx, int hc, int H, int W, int C, int oH, int oW,
                       const int* __restrict__ left_r, const float* __restrict__ dis_r,
                       const int* __restrict__ left_c, const float* __restrict__ dis_c,
                       const double* __restrict__ dis_r64, const double* __restrict__ dis_c64,
                       float max_sigma, uint8_t* __restrict__ out, int64_t oy, int64_t ox, int64_t oc, int pad_mode) {
#pragma clang fp contract(off)
    constexpr int S = 2;
    constexpr bool GAUSS = KIND == LERF_KIND_GAUSS;
    const int xq = blockIdx.x * 256 + threadIdx.x;
    if (xq >= (W + S) * C) return;
    const int qc = xq / C, c = xq - qc * C;
    const int lr = (int)blockIdx.y - S, lc = qc - S;                  // the cell's first tap: -S .. H - 1, -S .. W - 1
    const float ry = (float)oH / (float)H, rx = (float)oW / (float)W;
    const int i0 = first_at_least(left_r, oH, lr, (int)((float)(lr + 1) * ry) - 1);
    const int i1 = first_at_least(left_r, oH, lr + 1, i0 + (int)ry);
    if (i0 >= i1) return;
    const int j0 = first_at_least(left_c, oW, lc, (int)((float)(lc + 1) * rx) - 1);
    const int j1 = first_at_least(left_c, oW, lc + 1, j0 + (int)rx);
    if (j0 >= j1) return;
    uint32_t dd[S * S];                                               // (k0, k1, k2, value) per tap: what the tie guard reads
    float v[S * S];
#pragma unroll
    for (int a = 0; a < S; ++a) {
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const int rr = lr + b, cc = lc + a;
            const int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);                          // hyper: edge pad (:172-174)
            bool zr, zc;                                                                               // image pad rule (:208)
            const int rs = pad_index(rr, H, pad_mode, &zr), cs = pad_index(cc, W, pad_mode, &zc);
            const uint32_t val = (zr || zc) ? 0u : (uint32_t)feat[rs * fy + cs * fx + c * fc];
            const int hh = rcl * hy + ccl * hx + c * hc;
            dd[a * S + b] = (uint32_t)h0[hh] | ((GAUSS ? (uint32_t)h1[hh] : 0u) << 8) | ((GAUSS ? (uint32_t)h2[hh] : 0u) << 16) | (val << 24);
            v[a * S + b] = (float)val;
        }
    }
    uint8_t* const oc0 = out + (int64_t)c * oc;
    auto put = [&](int i, int j, float xf) {
        uint8_t r;
        if (dis_r64 != nullptr && s3::near_tie(xf)) {
            double dx64[S], dy64[S];
#pragma unroll
            for (int b = 0; b < S; ++b) dx64[b] = dis_r64[i * S + b];
#pragma unroll
            for (int a = 0; a < S; ++a) dy64[a] = dis_c64[j * S + a];
            r = s3::resolve_u8<GAUSS, S>(dd, dx64, dy64, max_sigma);
        } else {
            r = s3::to_u8(xf);
        }
        oc0[(int64_t)i * oy + (int64_t)j * ox] = r;
    };
    if (GAUSS) {
        const float gs = s3::gauss_scale(max_sigma);
        float m2rho[S * S], k1[S * S], k2[S * S];
#pragma unroll
        for (int t = 0; t < S * S; ++t) {
            m2rho[t] = s3::gauss_m2rho_u8((float)(dd[t] & 0xFFu));
            k1[t] = (float)((dd[t] >> 8) & 0xFFu);
            k2[t] = (float)((dd[t] >> 16) & 0xFFu);
        }
        for (int j = j0; j < j1; ++j) {
            float ty2[S * S], mty[S * S];
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float dys = dis_c[j * S + a] * gs;
#pragma unroll
                for (int b = 0; b < S; ++b) {
                    const float ty = s3::gauss_t_u8(k2[a * S + b], dys);
                    ty2[a * S + b] = ty * ty;
                    mty[a * S + b] = m2rho[a * S + b] * ty;
                }
            }
            for (int i = i0; i < i1; ++i) {
                float e[S * S];
#pragma unroll
                for (int b = 0; b < S; ++b) {
                    const float dxs = dis_r[i * S + b] * gs;
#pragma unroll
                    for (int a = 0; a < S; ++a)
                        e[a * S + b] = s3::gauss_form_cols(s3::gauss_t_u8(k1[a * S + b], dxs), ty2[a * S + b], mty[a * S + b]);
                }
                put(i, j, s3::finish<true, S * S, true, true, true>(e, v));
            }
        }
    } else {
        const float ms255 = max_sigma * (1.0f / 255.0f);
        float alpha[S * S];
#pragma unroll
        for (int t = 0; t < S * S; ++t) alpha[t] = s3::lin_alpha_u8((float)(dd[t] & 0xFFu), ms255);
        for (int j = j0; j < j1; ++j) {
            float fyv[S * S];
#pragma unroll
            for (int a = 0; a < S; ++a) {
                const float dy = dis_c[j * S + a];
                const int cls = s3::dist_class_f(dy);
#pragma unroll
                for (int b = 0; b < S; ++b) fyv[a * S + b] = s3::lin_factor(alpha[a * S + b], dy, cls);
            }
            for (int i = i0; i < i1; ++i) {
                float e[S * S];
#pragma unroll
                for (int b = 0; b < S; ++b) {
                    const float dx = dis_r[i * S + b];
                    const int cls = s3::dist_class_f(dx);
#pragma unroll
                    for (int a = 0; a < S; ++a) e[a * S + b] = s3::lin_factor(alpha[a * S + b], dx, cls) * fyv[a * S + b];
                }
                put(i, j, s3::finish<false, S * S, true, true, true>(e, v));
            }
        }
    }
}

// the cell kernel takes the call when every operand is uint8, S = 2, the grid is an up-sampling one (the cells' outputs are
// contiguous runs of the tables) and the planes can be addressed with 32-bit offsets
template <int KIND>
static bool resize_cells_u8(const ResizeArgs& a, hipStream_t st) {
    if (a.S != 2 || a.oH < a.H || a.oW < a.W || !a.dis_r || !a.dis_c || !a.left_r || !a.left_c) return false;
    auto fits = [](int64_t sy, int64_t sx, int64_t sc, int H, int W, int C) {
        const int64_t m = (sy < 0 ? -sy : sy) * (H - 1) + (sx < 0 ? -sx : sx) * (W - 1) + (sc < 0 ? -sc : sc) * (C - 1);
        return sy >= 0 && sx >= 0 && sc >= 0 && m < (1ll << 31) - 1;
    };
    if (!fits(a.fy, a.fx, a.fc, a.H, a.W, a.C) || !fits(a.hy, a.hx, a.hc, a.H, a.W, a.C)) return false;
    const int64_t xq = (int64_t)(a.W + 2) * a.C;
    if (xq > 0x7FFFFFFF || (int64_t)a.H + 2 > 65535) return false;
    dim3 block(256), grid((unsigned)((xq + 255) / 256), (unsigned)(a.H + 2));
    const bool guard = a.dis_r64 && a.dis_c64;
    hipLaunchKernelGGL((resize_cells_u8_kernel<KIND>), grid, block, 0, st, (const uint8_t*)a.feat, (int)a.fy, (int)a.fx, (int)a.fc,
                       (const uint8_t*)a.h[0], (const uint8_t*)a.h[1], (const uint8_t*)a.h[2], (int)a.hy, (int)a.hx, (int)a.hc,
                       a.H, a.W, a.C, a.oH, a.oW, a.left_r, a.dis_r, a.left_c, a.dis_c, guard ? a.dis_r64 : nullptr,
                       guard ? a.dis_c64 : nullptr, (float)a.max_sigma, (uint8_t*)a.out, a.oy, a.ox, a.oc, a.pad_mode);
    return true;
}

template <typename TI, typename TH, typename TO, typename A, int KIND>
static int resize_dispatch_S(const ResizeArgs& a, hipStream_t st) {
#ifndef LERF_NO_RESIZE_CELLS                                          // A/B builds only (tools/build_li_variant.sh): never the product
    if constexpr (sizeof(TI) == 1 && sizeof(TH) == 1 && sizeof(TO) == 1 && sizeof(A) == 4) {
        if (resize_cells_u8<KIND>(a, st)) return LERF_OK;
    }
#endif
    dim3 block(256), grid((a.oW * a.C + 255) / 256, a.oH);
    const A* dr = sizeof(A) == 4 ? (const A*)a.dis_r : (const A*)a.dis_r64;
    const A* dc = sizeof(A) == 4 ? (const A*)a.dis_c : (const A*)a.dis_c64;
    if (!dr || !dc) return LERF_EINVAL;
#define LERF_RS(ST)                                                                                         \
    hipLaunchKernelGGL((resize_kernel<TI, TH, TO, A, KIND, ST>), grid, block, 0, st, (const TI*)a.feat,     \
                       a.fy, a.fx, a.fc, (const TH*)a.h[0], (const TH*)a.h[1], (const TH*)a.h[2], a.hy,     \
                       a.hx, a.hc, a.H, a.W, a.C, a.S, a.oH, a.oW, a.left_r, dr, a.left_c, dc,              \
                       (a.dis_r64 && a.dis_c64) ? a.dis_r64 : nullptr, (a.dis_r64 && a.dis_c64) ? a.dis_c64 : nullptr, \
                       (A)a.max_sigma, (TO*)a.out, a.oy, a.ox, a.oc, a.pad_mode)
    if (a.S == 2) LERF_RS(2);
    else if (a.S == 4) LERF_RS(4);
    else LERF_RS(0);
#undef LERF_RS
    return LERF_OK;
}

template <typename TI, typename TH, typename TO, typename A>
static int resize_dispatch_kind(const ResizeArgs& a, hipStream_t st) {
    if (a.kind == LERF_KIND_GAUSS) return resize_dispatch_S<TI, TH, TO, A, LERF_KIND_GAUSS>(a, st);
    if (a.kind == LERF_KIND_LINEAR) return resize_dispatch_S<TI, TH, TO, A, LERF_KIND_LINEAR>(a, st);
    return LERF_EUNSUPPORTED;
}

static int launch_resize_fixed(const ResizeArgs& a, hipStream_t st);

int launch_resize(const ResizeArgs& a, hipStream_t st) {
    if (a.S < 1 || a.S > LERF_MAX_SUPPORT) return LERF_EUNSUPPORTED;
    if (a.kind >= LERF_KIND_NEAREST && a.kind <= LERF_KIND_LANCZOS3) return launch_resize_fixed(a, st);
    if (a.in_dtype == LERF_U8 && a.h_dtype == LERF_U8) {
        if (a.out_dtype == LERF_U8) {
            // float32 production arithmetic + tie guard, sized for max_sigma <= 13 (lerf_stage3.h); beyond that the
            // reference's own float64 arithmetic, rounded once
            if (a.kind == LERF_KIND_GAUSS && a.max_sigma > (double)s3::kNoShiftMaxSigma)
                return resize_dispatch_kind<uint8_t, uint8_t, uint8_t, double>(a, st);
            return resize_dispatch_kind<uint8_t, uint8_t, uint8_t, float>(a, st);
        }
        if (a.out_dtype == LERF_F32) return resize_dispatch_kind<uint8_t, uint8_t, float, double>(a, st);   // float64 arithmetic, rounded once
        if (a.out_dtype == LERF_F64) return resize_dispatch_kind<uint8_t, uint8_t, double, double>(a, st);
    } else if (a.in_dtype == LERF_F32 && a.h_dtype == LERF_F32) {
        if (a.out_dtype == LERF_F32) return resize_dispatch_kind<float, float, float, double>(a, st);
        if (a.out_dtype == LERF_F64) return resize_dispatch_kind<float, float, double, double>(a, st);
    }
    return LERF_EUNSUPPORTED;
}

// fixed interpolation kernels of resize_right/interp_methods.py:35-70 (the reference's non-learned warps,
// resize_right2d_numpy.py:451-494); evaluated in float64 like the reference, eps = float32 eps
__device__ __forceinline__ double fixed_kernel_1d(int kind, double x) {
#pragma clang fp contract(off)
    const double pi = 3.141592653589793;
    const double eps = (double)kEps32;
    if (kind == LERF_KIND_CUBIC) {                                        // :35-43
        const double a = fabs(x), a2 = a * a, a3 = a * a * a;
        return (1.5 * a3 - 2.5 * a2 + 1.0) * (a <= 1.0 ? 1.0 : 0.0) +
               (-0.5 * a3 + 2.5 * a2 - 4.0 * a + 2.0) * ((1.0 < a && a <= 2.0) ? 1.0 : 0.0);
    }
    if (kind == LERF_KIND_LANCZOS2)                                       // :46-50
        return ((sin(pi * x) * sin(pi * x / 2) + eps) / ((pi * pi * (x * x) / 2) + eps)) * (fabs(x) < 2.0 ? 1.0 : 0.0);
    if (kind == LERF_KIND_LANCZOS3)                                       // :53-57
        return ((sin(pi * x) * sin(pi * x / 3) + eps) / ((pi * pi * (x * x) / 3) + eps)) * (fabs(x) < 3.0 ? 1.0 : 0.0);
    if (kind == LERF_KIND_BILINEAR)                                       // :60-64
        return (x + 1.0) * ((-1.0 <= x && x < 0.0) ? 1.0 : 0.0) + (1.0 - x) * ((0.0 <= x && x <= 1.0) ? 1.0 : 0.0);
    return ((-1.0 <= x && x < 0.0) ? 1.0 : 0.0) + ((0.0 <= x && x <= 1.0) ? 1.0 : 0.0);   // box :67-70
}

// ---------------------------------------------------------------------------
// Fixed-kernel SR: Resize2dTorch.resize + BicubicResize2dTorch (resize_right2d_torch.py:105-138)
// and its bilinear / lanczos / box siblings on the separable SR geometry.  weight(dx, dy) = k(dx) k(dy), normalised
// over the S x S patch (the sum factorises into a row sum times a column sum); zero-padded image; S = 1 is not
// normalised (:119-121).
// ---------------------------------------------------------------------------
template <typename TI, typename TO, typename A>
__global__ void __launch_bounds__(256)
resize_fixed_kernel(const TI* __restrict__ feat, int64_t fy, int64_t fx, int64_t fc, int H, int W, int C, int S, int oH, int oW,
                    const int* __restrict__ left_r, const A* __restrict__ dis_r, const int* __restrict__ left_c,
                    const A* __restrict__ dis_c, int kind, TO* __restrict__ out, int64_t oy, int64_t ox, int64_t oc, int pad_mode) {
    int xc = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (xc >= oW * C) return;
    int j = xc / C;
    int c = xc - j * C;
    const int lr = left_r[i], lc = left_c[j];
    A kr[LERF_MAX_SUPPORT], kc[LERF_MAX_SUPPORT], sr = 0, sc = 0;
    for (int b = 0; b < S; ++b) {
        kr[b] = (A)fixed_kernel_1d(kind, (double)dis_r[i * S + b]);
        kc[b] = (A)fixed_kernel_1d(kind, (double)dis_c[j * S + b]);
        sr += kr[b];
        sc += kc[b];
    }
    A num = 0;
    for (int a = 0; a < S; ++a) {
        bool zc;
        const int cc = pad_index(lc + a, W, pad_mode, &zc);
        if (zc) continue;
        A row = 0;
        for (int b = 0; b < S; ++b) {
            bool zr;
            const int rr = pad_index(lr + b, H, pad_mode, &zr);
            if (!zr) row += kr[b] * (A)Loader<TI>::pixel(feat + rr * fy + cc * fx + c * fc);
        }
        num += kc[a] * row;
    }
    Storer<TO>::put(out + i * oy + j * ox + c * oc, S == 1 ? num : num / (sr * sc));
}

template <typename TI, typename TO, typename A>
static int resize_fixed_launch(const ResizeArgs& a, hipStream_t st) {
    dim3 block(256), grid((a.oW * a.C + 255) / 256, a.oH);
    const A* dr = sizeof(A) == 4 ? (const A*)a.dis_r : (const A*)a.dis_r64;
    const A* dc = sizeof(A) == 4 ? (const A*)a.dis_c : (const A*)a.dis_c64;
    if (!dr || !dc) return LERF_EINVAL;
    hipLaunchKernelGGL((resize_fixed_kernel<TI, TO, A>), grid, block, 0, st, (const TI*)a.feat, a.fy, a.fx, a.fc, a.H, a.W, a.C,
                       a.S, a.oH, a.oW, a.left_r, dr, a.left_c, dc, a.kind, (TO*)a.out, a.oy, a.ox, a.oc, a.pad_mode);
    return LERF_OK;
}

static int launch_resize_fixed(const ResizeArgs& a, hipStream_t st) {
    if (a.in_dtype == LERF_U8) {
        if (a.out_dtype == LERF_U8) return resize_fixed_launch<uint8_t, uint8_t, float>(a, st);
        if (a.out_dtype == LERF_F32) return resize_fixed_launch<uint8_t, float, float>(a, st);
        if (a.out_dtype == LERF_F64) return resize_fixed_launch<uint8_t, double, double>(a, st);
    } else if (a.in_dtype == LERF_F32) {
        if (a.out_dtype == LERF_F32) return resize_fixed_launch<float, float, float>(a, st);
        if (a.out_dtype == LERF_F64) return resize_fixed_launch<float, double, double>(a, st);
    }
    return LERF_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------
// A7/A8: homographic warp, geometry per output pixel in float64
// ---------------------------------------------------------------------------
template <typename TI, typename TH, typename TO, typename A, int KIND>
__global__ void __launch_bounds__(256)
warp_kernel(const TI* __restrict__ feat, int64_t fy, int64_t fx, int64_t fc,
            const TH* __restrict__ h0, const TH* __restrict__ h1, const TH* __restrict__ h2,
            int64_t hy, int64_t hx, int64_t hc, int H, int W, int C, WarpGeo g,
            A max_sigma, TO* __restrict__ out, int64_t oy, int64_t ox, int64_t oc) {
    int xc = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (xc >= g.oW * C) return;
    int j = xc / C;
    int c = xc - j * C;
    const int S = g.S;
    double gr, gc;
    project_point(g.minv, i + g.oy0, j + g.ox0, H, W, &gr, &gc);
    int lr = left_boundary(gr, S) + g.pad_r_lo;
    int lc = left_boundary(gc, S) + g.pad_c_lo;
    gr += (double)g.pad_r_lo;      // calc_pad_sz shifts grid and field of view (:366-367)
    gc += (double)g.pad_c_lo;
    A emin = 0, num = 0, den = 0;
    for (int pass = (KIND == LERF_KIND_GAUSS ? 0 : 1); pass < 2; ++pass) {
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                // field of view clipped to [0, in-1] while indexing the PADDED arrays (:396-398)
                int pr = clampi(lr + b, 0, H - 1), pc = clampi(lc + a, 0, W - 1);
                double dxd = gr - (double)pr, dyd = gc - (double)pc;
                A dx = (A)dxd, dy = (A)dyd;
                int sr = pr - g.pad_r_lo, sc_ = pc - g.pad_c_lo;       // unpadded source coordinates
                int rcl = clampi(sr, 0, H - 1), ccl = clampi(sc_, 0, W - 1);
                bool zr, zc;                                           // image pad rule (:560)
                const int rs = pad_index(sr, H, g.pad_mode, &zr), cs = pad_index(sc_, W, g.pad_mode, &zc);
                int64_t ho = rcl * hy + ccl * hx + c * hc;
                A w;
                if (KIND == LERF_KIND_GAUSS) {
                    float p0 = Loader<TH>::hyper(h0 + ho), p1 = Loader<TH>::hyper(h1 + ho), p2 = Loader<TH>::hyper(h2 + ho);
                    A rho = (A)(p0 * 2.0f - 1.0f), sx = (A)(p1 * (float)max_sigma), sy = (A)(p2 * (float)max_sigma);
                    A tx = sx * dx, ty = sy * dy;
                    A e = tx * tx - (A)2 * rho * (tx * ty) + ty * ty;
                    if (pass == 0) {
                        emin = sizeof(A) == 8 ? (A)0 : ((a == 0 && b == 0) ? e : (e < emin ? e : emin));   // float64: unshifted, as TapAcc
                        continue;
                    }
                    w = sizeof(A) == 4 ? (A)__expf((float)((A)-0.5 * (e - emin))) : (A)exp((double)((A)-0.5 * (e - emin)));
                } else if (KIND == LERF_KIND_LINEAR) {
                    float p0 = Loader<TH>::hyper(h0 + ho);
                    A alpha = (A)((float)max_sigma * (p0 * 2.0f - 1.0f));
                    // class decisions on the float64 distances
                    w = lin_factor<A>(alpha, dx, dist_class(dxd)) * lin_factor<A>(alpha, dy, dist_class(dyd));
                } else if (KIND == LERF_KIND_NEAREST) {
                    w = (dist_class(dxd) != 0 && dist_class(dyd) != 0) ? (A)1 : (A)0;     // box2d
                } else {
                    w = (A)(fixed_kernel_1d(KIND, dxd) * fixed_kernel_1d(KIND, dyd));     // cubic2d / linear2d / lanczos
                }
                A val = (zr || zc) ? (A)0 : (A)Loader<TI>::pixel(feat + rs * fy + cs * fx + c * fc);
                num += w * val;
                den += w;
            }
    }
    A res = num / den;
    if (KIND == LERF_KIND_GAUSS && emin * (A)0.5 > (A)745.2) res = (A)(0.0 / 0.0);
    if (sizeof(TI) == 1 && sizeof(TH) == 1 && sizeof(TO) == 1) {
        auto tap = [&](int rcl, int ccl) -> uint32_t {
            const int64_t ho = rcl * hy + ccl * hx + c * hc;
            const uint32_t k0 = (uint32_t)h0[ho];
            const uint32_t k12 = KIND == LERF_KIND_GAUSS ? (((uint32_t)h1[ho]) << 8) | (((uint32_t)h2[ho]) << 16) : 0u;
            return k0 | k12 | ((uint32_t)feat[rcl * fy + ccl * fx + c * fc] << 24);
        };
        if (warp_tie_guard<KIND>((float)res, S, H, W, g, lr, lc, gr, gc, (float)max_sigma, tap,
                                 reinterpret_cast<uint8_t*>(out + i * oy + j * ox + c * oc)))
            return;
    }
    Storer<TO>::put(out + i * oy + j * ox + c * oc, res);
}

template <typename TI, typename TH, typename TO, typename A>
static int warp_dispatch_kind(const WarpArgs& a, hipStream_t st) {
    dim3 block(256), grid((a.geo.oW * a.C + 255) / 256, a.geo.oH);
#define LERF_WP(KIND)                                                                                        \
    hipLaunchKernelGGL((warp_kernel<TI, TH, TO, A, KIND>), grid, block, 0, st, (const TI*)a.feat, a.fy,      \
                       a.fx, a.fc, (const TH*)a.h[0], (const TH*)a.h[1], (const TH*)a.h[2], a.hy, a.hx,      \
                       a.hc, a.H, a.W, a.C, a.geo, (A)a.max_sigma, (TO*)a.out, a.oy, a.ox, a.oc)
    if (a.kind == LERF_KIND_GAUSS) LERF_WP(LERF_KIND_GAUSS);
    else if (a.kind == LERF_KIND_LINEAR) LERF_WP(LERF_KIND_LINEAR);
    else if (a.kind == LERF_KIND_NEAREST) LERF_WP(LERF_KIND_NEAREST);
    else if (a.kind == LERF_KIND_CUBIC) LERF_WP(LERF_KIND_CUBIC);
    else if (a.kind == LERF_KIND_BILINEAR) LERF_WP(LERF_KIND_BILINEAR);
    else if (a.kind == LERF_KIND_LANCZOS2) LERF_WP(LERF_KIND_LANCZOS2);
    else if (a.kind == LERF_KIND_LANCZOS3) LERF_WP(LERF_KIND_LANCZOS3);
    else return LERF_EUNSUPPORTED;
#undef LERF_WP
    return LERF_OK;
}

int launch_warp(const WarpArgs& a, hipStream_t st) {
    if (a.geo.S < 1 || a.geo.S > LERF_MAX_SUPPORT) return LERF_EUNSUPPORTED;
    const bool fixed = a.kind >= LERF_KIND_NEAREST;     // no hyper-parameter maps
    if (a.in_dtype == LERF_U8 && (a.h_dtype == LERF_U8 || fixed)) {
        if (a.out_dtype == LERF_U8) return warp_dispatch_kind<uint8_t, uint8_t, uint8_t, float>(a, st);
        if (a.out_dtype == LERF_F32) return warp_dispatch_kind<uint8_t, uint8_t, float, double>(a, st);     // float64 arithmetic, rounded once
        if (a.out_dtype == LERF_F64) return warp_dispatch_kind<uint8_t, uint8_t, double, double>(a, st);
    } else if (a.in_dtype == LERF_F32 && (a.h_dtype == LERF_F32 || fixed)) {
        if (a.out_dtype == LERF_F32) return warp_dispatch_kind<float, float, float, double>(a, st);
        if (a.out_dtype == LERF_F64) return warp_dispatch_kind<float, float, double, double>(a, st);
    }
    return LERF_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------
// packed stage outputs (tile-fused stages kernel): dword = hq0 | hq1<<8 | hq2<<16 | feat<<24
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
unpack_stages_kernel(const uint32_t* __restrict__ packed, int64_t n, int oC, uint8_t* __restrict__ feat,
                     uint8_t* __restrict__ hq) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = packed[i];
    if (feat) feat[i] = (uint8_t)(d >> 24);
    if (hq)
        for (int k = 0; k < oC; ++k) hq[i * oC + k] = (uint8_t)(d >> (8 * k));
}

int launch_unpack_stages(const uint32_t* packed, int64_t n_pxch, int oC, uint8_t* feat, uint8_t* hq, hipStream_t st) {
    if (oC != 1 && oC != 3) return LERF_EUNSUPPORTED;
    hipLaunchKernelGGL(unpack_stages_kernel, dim3((unsigned)((n_pxch + 255) / 256)), dim3(256), 0, st, packed, n_pxch, oC,
                       feat, hq);
    return LERF_OK;
}

// A7/A8 on packed stage outputs: one dword load per tap (the warp harness path,
// resample/eval_lut_warp.py:100-222 with stages 1+2 from the fused kernel)
template <typename TO, int KIND>
__global__ void __launch_bounds__(256)
warp_packed_kernel(const uint32_t* __restrict__ packed, int64_t packed_sn, int H, int W, int C, WarpGeo g, float max_sigma,
                   TO* __restrict__ out, int64_t oy, int64_t ox, int64_t oc, int64_t out_sn) {
    packed += (int64_t)blockIdx.z * packed_sn;             // frame of the batch (one homography for all)
    out += (int64_t)blockIdx.z * out_sn;
    int xc = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (xc >= g.oW * C) return;
    int j = xc / C;
    int c = xc - j * C;
    const int S = g.S;
    double gr, gc;
    project_point(g.minv, i + g.oy0, j + g.ox0, H, W, &gr, &gc);
    int lr = left_boundary(gr, S) + g.pad_r_lo;
    int lc = left_boundary(gc, S) + g.pad_c_lo;
    gr += (double)g.pad_r_lo;
    gc += (double)g.pad_c_lo;
    float emin = 0, num = 0, den = 0;
    for (int pass = (KIND == LERF_KIND_GAUSS ? 0 : 1); pass < 2; ++pass) {
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                int pr = clampi(lr + b, 0, H - 1), pc = clampi(lc + a, 0, W - 1);       // :396-398
                double dxd = gr - (double)pr, dyd = gc - (double)pc;
                float dx = (float)dxd, dy = (float)dyd;
                int sr = pr - g.pad_r_lo, sc_ = pc - g.pad_c_lo;
                int rcl = clampi(sr, 0, H - 1), ccl = clampi(sc_, 0, W - 1);
                bool inside = (sr == rcl) && (sc_ == ccl);
                const uint32_t d = packed[((int64_t)rcl * W + ccl) * C + c];
                float w;
                if (KIND == LERF_KIND_GAUSS) {
                    float e = s3::gauss_form(s3::u8_over_255((float)(d & 0xFFu)), s3::u8_over_255((float)((d >> 8) & 0xFFu)),
                                             s3::u8_over_255((float)((d >> 16) & 0xFFu)), max_sigma, dx, dy);
                    if (pass == 0) {
                        emin = (a == 0 && b == 0) ? e : fminf(e, emin);
                        continue;
                    }
                    w = s3::gauss_weight(e, emin);
                } else {
                    float alpha = s3::lin_alpha_of(s3::u8_over_255((float)(d & 0xFFu)), max_sigma);
                    w = s3::lin_factor(alpha, dx, dist_class(dxd)) * s3::lin_factor(alpha, dy, dist_class(dyd));
                }
                float val = inside ? (float)(d >> 24) : 0.0f;
                num += w * val;
                den += w;
            }
    }
    float res = num / den;
    if (KIND == LERF_KIND_GAUSS && emin * 0.5f > 745.2f) res = __builtin_nanf("");
    if (sizeof(TO) == 1) {
        auto tap = [&](int rcl, int ccl) -> uint32_t { return packed[((int64_t)rcl * W + ccl) * C + c]; };
        if (warp_tie_guard<KIND>(res, S, H, W, g, lr, lc, gr, gc, max_sigma, tap,
                                 reinterpret_cast<uint8_t*>(out + i * oy + j * ox + c * oc)))
            return;
    }
    Storer<TO>::put(out + i * oy + j * ox + c * oc, res);
}

// The same, one thread per output PIXEL of an RGB frame with S = 2: the float64 projection, the support and the
// four tap positions / distances are computed once and shared by the three channels (the per-channel kernel above
// spends most of its time repeating the two float64 divisions of the projection).
// PROD (uint8 outputs, max_sigma <= s3::kNoShiftMaxSigma): the float32 production arithmetic of the SR kernels -- forms
// pre-scaled by 0.5 log2(e) straight from the uint8 numerators (lerf_stage3.h gauss_form_u8), exp2 weights shifted by the
// support's minimum, reciprocal + Newton step -- instead of the exact float32 parameter formation: 75 instead of 140 VALU
// instructions per output value, same bytes (the tie guard re-evaluates anything within 1.5e-4 of a rounding tie in float64).
template <typename TO, int KIND, bool PROD = false>
__global__ void __launch_bounds__(256)
warp_packed_px_kernel(const uint32_t* __restrict__ packed0, int64_t packed_sn, int n_frames, int H, int W, WarpGeo g, float max_sigma,
                      TO* __restrict__ out0, int64_t oy, int64_t ox, int64_t oc, int64_t out_sn) {
    constexpr int S = 2, C = 3;
    // 1-D grid of (output row, 256-pixel segment) blocks in row-major order, each XCD on a contiguous eighth of it = a band of
    // output rows: the packed-map rows two neighbouring output rows share are then fetched into ONE L2 (round 4, linear
    // order: every map byte came from HBM 2.5 times, L2 hit 0.62 -- neighbouring rows sat on different XCDs)
    const int gx = (g.oW + 255) >> 8;
    const int b = (int)gridDim.x >= 1024 ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int i = b / gx;
    const int j = (b - i * gx) * 256 + (int)threadIdx.x;
    if (j >= g.oW) return;
    const WarpPx2 G = warp_px_geometry(g, i, j, H, W);
    const int lr = G.lr, lc = G.lc;
    const double gr = G.gr, gc = G.gc;
    int64_t pos[S * S];
    float dx[S], dy[S];
    int cx[S], cy[S];
    bool in_r[S], in_c[S];
#pragma unroll
    for (int b = 0; b < S; ++b) { dx[b] = G.dx[b]; cx[b] = G.cx[b]; in_r[b] = G.in_r[b]; dy[b] = G.dy[b]; cy[b] = G.cy[b]; in_c[b] = G.in_c[b]; }
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) pos[a * S + b] = ((int64_t)G.rrow[b] * W + G.rcol[a]) * C;
    float dxs[S], dys[S];
    const float gsc = (PROD && KIND == LERF_KIND_GAUSS) ? s3::gauss_scale(max_sigma) : 1.0f;
    const float ms255 = max_sigma * (1.0f / 255.0f);
#pragma unroll
    for (int b = 0; b < S; ++b) { dxs[b] = dx[b] * gsc; dys[b] = dy[b] * gsc; }
    // the frames of the batch share the homography: the float64 projection and the tap geometry above are paid once per
    // output pixel, not once per frame (round 3: a fifth of this kernel's instructions went into repeating them)
#pragma unroll 1
    for (int fr = 0; fr < n_frames; ++fr) {
    const uint32_t* __restrict__ packed = packed0 + (int64_t)fr * packed_sn;
    TO* __restrict__ out = out0 + (int64_t)fr * out_sn;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        uint32_t d[S * S];
#pragma unroll
        for (int t = 0; t < S * S; ++t) d[t] = packed[pos[t] + c];
        if constexpr (PROD) {
            static_assert(sizeof(TO) == 1, "production arithmetic: uint8 outputs");
            TO* dst = out + i * oy + j * ox + c * oc;
            auto tap = [&](int rcl, int ccl) -> uint32_t { return packed[((int64_t)rcl * W + ccl) * C + c]; };
            float res;
            if (warp_px_value_u8<KIND>(G, g, H, W, max_sigma, dxs, dys, tap, reinterpret_cast<uint8_t*>(dst), &res)) continue;
            Storer<TO>::put(dst, res);
            continue;
        }
        float e[S * S], emin = 0.0f, num = 0.0f, den = 0.0f;
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) {
                const uint32_t q = d[a * S + b];
                if (KIND == LERF_KIND_GAUSS) {
                    e[a * S + b] = s3::gauss_form(s3::u8_over_255((float)(q & 0xFFu)), s3::u8_over_255((float)((q >> 8) & 0xFFu)),
                                                  s3::u8_over_255((float)((q >> 16) & 0xFFu)), max_sigma, dx[b], dy[a]);
                    emin = (a == 0 && b == 0) ? e[0] : fminf(e[a * S + b], emin);
                } else {
                    const float alpha = s3::lin_alpha_of(s3::u8_over_255((float)(q & 0xFFu)), max_sigma);
                    e[a * S + b] = s3::lin_factor(alpha, dx[b], cx[b]) * s3::lin_factor(alpha, dy[a], cy[a]);
                }
            }
#pragma unroll
        for (int a = 0; a < S; ++a)
#pragma unroll
            for (int b = 0; b < S; ++b) {
                const float w = KIND == LERF_KIND_GAUSS ? s3::gauss_weight(e[a * S + b], emin) : e[a * S + b];
                const float val = (in_r[b] && in_c[a]) ? (float)(d[a * S + b] >> 24) : 0.0f;
                num += w * val;
                den += w;
            }
        float res = num / den;
        if (KIND == LERF_KIND_GAUSS && emin * 0.5f > 745.2f) res = __builtin_nanf("");
        TO* dst = out + i * oy + j * ox + c * oc;
        if (sizeof(TO) == 1) {
            auto tap = [&](int rcl, int ccl) -> uint32_t { return packed[((int64_t)rcl * W + ccl) * C + c]; };
            if (warp_tie_guard<KIND>(res, S, H, W, g, lr, lc, gr, gc, max_sigma, tap, reinterpret_cast<uint8_t*>(dst))) continue;
        }
        Storer<TO>::put(dst, res);
    }
    }
}

int launch_warp_packed(const uint32_t* packed, int64_t packed_sn, int n, int H, int W, int C, const WarpGeo& geo, int kind,
                       float max_sigma, void* out, int out_dtype, int64_t oy, int64_t ox, int64_t oc, int64_t out_sn, hipStream_t st) {
    if (geo.S < 1 || geo.S > LERF_MAX_SUPPORT) return LERF_EUNSUPPORTED;
    if (n < 1 || n > 65535 || geo.oH > 65535) return LERF_EUNSUPPORTED;
    if (C == 3 && geo.S == 2 && (kind == LERF_KIND_GAUSS || kind == LERF_KIND_LINEAR) &&
        (out_dtype == LERF_U8 || out_dtype == LERF_F32)) {
        dim3 blockp(256), gridp((unsigned)(((geo.oW + 255) / 256) * geo.oH), 1, 1);
#define LERF_WPX(TO, KIND, PROD)                                                                                      \
    hipLaunchKernelGGL((warp_packed_px_kernel<TO, KIND, PROD>), gridp, blockp, 0, st, packed, packed_sn, n, H, W, geo, max_sigma, (TO*)out, \
                       oy, ox, oc, out_sn)
        const bool prod = out_dtype == LERF_U8 && max_sigma <= s3::kNoShiftMaxSigma;     // production arithmetic + tie guard
        if (kind == LERF_KIND_GAUSS) {
            if (prod) LERF_WPX(uint8_t, LERF_KIND_GAUSS, true);
            else if (out_dtype == LERF_U8) LERF_WPX(uint8_t, LERF_KIND_GAUSS, false);
            else LERF_WPX(float, LERF_KIND_GAUSS, false);
        } else {
            if (prod) LERF_WPX(uint8_t, LERF_KIND_LINEAR, true);
            else if (out_dtype == LERF_U8) LERF_WPX(uint8_t, LERF_KIND_LINEAR, false);
            else LERF_WPX(float, LERF_KIND_LINEAR, false);
        }
#undef LERF_WPX
        return LERF_OK;
    }
    dim3 block(256), grid((geo.oW * C + 255) / 256, geo.oH, n);
#define LERF_WPK(TO, KIND)                                                                                         \
    hipLaunchKernelGGL((warp_packed_kernel<TO, KIND>), grid, block, 0, st, packed, packed_sn, H, W, C, geo, max_sigma, (TO*)out, \
                       oy, ox, oc, out_sn)
    if (kind == LERF_KIND_GAUSS) {
        if (out_dtype == LERF_U8) LERF_WPK(uint8_t, LERF_KIND_GAUSS);
        else if (out_dtype == LERF_F32) LERF_WPK(float, LERF_KIND_GAUSS);
        else return LERF_EUNSUPPORTED;
    } else if (kind == LERF_KIND_LINEAR) {
        if (out_dtype == LERF_U8) LERF_WPK(uint8_t, LERF_KIND_LINEAR);
        else if (out_dtype == LERF_F32) LERF_WPK(float, LERF_KIND_LINEAR);
        else return LERF_EUNSUPPORTED;
    } else {
        return LERF_EUNSUPPORTED;
    }
#undef LERF_WPK
    return LERF_OK;
}

// ---------------------------------------------------------------------------
// halo plumbing of the multi-GPU partitions: up to LERF_MAX_RECTS rectangles of a batch of dense uint8 frames <-> one
// contiguous staging buffer, ONE launch (lerf_rect_copy_u8).  Bandwidth-trivial (a few hundred KB); what matters is the
// launch count on a step of a few hundred microseconds.
// ---------------------------------------------------------------------------
struct RectSet {
    int n;
    int y[LERF_MAX_RECTS], x[LERF_MAX_RECTS], h[LERF_MAX_RECTS], wb[LERF_MAX_RECTS];   // wb = row bytes (w * C)
    int64_t off[LERF_MAX_RECTS];
    int64_t first[LERF_MAX_RECTS + 1];          // prefix sums of the rectangles' bytes over the batch
};
__global__ void __launch_bounds__(256)
rect_copy_kernel(uint8_t* __restrict__ frames, int n_frames, int fh, int fw, int C, uint8_t* __restrict__ staging, RectSet R,
                 int to_staging) {
    const int64_t total = R.first[R.n];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < LERF_MAX_RECTS; ++k)
            if (k < R.n && i >= R.first[k]) r = k;
        const int64_t e = i - R.first[r];                     // byte inside rectangle r's [n][h][w*C] block
        const int64_t per = (int64_t)R.h[r] * R.wb[r];
        const int f = (int)(e / per);
        const int64_t q = e - (int64_t)f * per;
        const int row = (int)(q / R.wb[r]), cb = (int)(q - (int64_t)row * R.wb[r]);
        uint8_t* fp = frames + ((int64_t)f * fh + R.y[r] + row) * ((int64_t)fw * C) + (int64_t)R.x[r] * C + cb;
        uint8_t* sp = staging + R.off[r] + e;
        if (to_staging) *sp = *fp; else *fp = *sp;
    }
}

int launch_rect_copy(uint8_t* frames, int n, int fh, int fw, int C, uint8_t* staging, const lerf_rect_t* rects, int n_rects,
                     int to_staging, hipStream_t st) {
    RectSet R{};
    R.n = n_rects;
    int64_t acc = 0;
    for (int r = 0; r < n_rects; ++r) {
        R.y[r] = rects[r].y; R.x[r] = rects[r].x; R.h[r] = rects[r].h; R.wb[r] = rects[r].w * C; R.off[r] = rects[r].off;
        R.first[r] = acc;
        acc += (int64_t)n * rects[r].h * rects[r].w * C;
    }
    for (int r = n_rects; r <= LERF_MAX_RECTS; ++r) R.first[r] = acc;
    const int64_t blocks = (acc + 255) / 256;
    hipLaunchKernelGGL(rect_copy_kernel, dim3((unsigned)(blocks < 4096 ? (blocks > 0 ? blocks : 1) : 4096)), dim3(256), 0, st, frames, n,
                       fh, fw, C, staging, R, to_staging);
    return LERF_OK;
}

}  // namespace lerf
