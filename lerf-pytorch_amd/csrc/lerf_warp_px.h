// One output pixel of a homographic warp (S = 2, RGB) on the PACKED stage outputs (hq0 | hq1 << 8 | hq2 << 16 | feat << 24 per
// pixel-channel): the float64 projection and tap geometry of Warp2dNumpy (resize_right/resize_right2d_numpy.py:306-407) and the
// float32 production arithmetic + float64 tie guard of the uint8 path.  Shared by warp_packed_px_kernel (taps from the packed
// maps in HBM / L2) and the tile-fused warp (taps from the tile's packed dwords in LDS, lerf_fused_impl.h): the same
// instructions in both, so fused == unfused bit for bit.
#pragma once

#include "lerf_kernels.h"
#include "lerf_stage3.h"

namespace lerf {

// amplified-linear 1-D factor (resize_right2d_numpy.py:233-241), cls = class of
// the float64 distance: 0 outside [-1,1], 1 for [-1,0), 2 for [0,1]
template <typename A>
__device__ __forceinline__ A lin_factor(A alpha, A x, int cls) {
    A f = cls == 1 ? alpha * x + (A)1 : (cls == 2 ? (A)1 - alpha * x : (A)0);
    return f < (A)0 ? (A)0 : f;
}
template <typename A>
__device__ __forceinline__ int dist_class(A x) {
    return (x >= (A)-1 && x < (A)0) ? 1 : ((x >= (A)0 && x <= (A)1) ? 2 : 0);
}

// Tie guard of the uint8 warps: an output within kTieEps of a half-integer is resolved in float64 (s3::resolve_u8: the
// reference's float64 forms, and its whole dtype chain where that is still undecided), like the SR kernels do; `tap(r, c)` returns (k0 | k1<<8 | k2<<16 | val<<24)
// of the clamped source pixel.
template <int KIND, int S, typename F>
__device__ __forceinline__ uint8_t warp_resolve_u8(int H, int W, const WarpGeo& g, int lr, int lc, double gr, double gc,
                                                   float max_sigma, F tap) {
    uint32_t dd[S * S];
    double dx[S], dy[S];
#pragma unroll
    for (int b = 0; b < S; ++b) dx[b] = gr - (double)clampi(lr + b, 0, H - 1);
#pragma unroll
    for (int a = 0; a < S; ++a) dy[a] = gc - (double)clampi(lc + a, 0, W - 1);
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const int sr = clampi(lr + b, 0, H - 1) - g.pad_r_lo, sc_ = clampi(lc + a, 0, W - 1) - g.pad_c_lo;
            const int rcl = clampi(sr, 0, H - 1), ccl = clampi(sc_, 0, W - 1);
            const uint32_t d = tap(rcl, ccl);
            dd[a * S + b] = ((sr == rcl) && (sc_ == ccl)) ? d : (d & 0x00FFFFFFu);      // zero image outside the frame
        }
    return s3::resolve_u8<KIND == LERF_KIND_GAUSS, S>(dd, dx, dy, max_sigma);
}

template <int KIND, typename F>
__device__ __forceinline__ bool warp_tie_guard(float res, int S, int H, int W, const WarpGeo& g, int lr, int lc, double gr,
                                               double gc, float max_sigma, F tap, uint8_t* dst) {
    if (!(KIND == LERF_KIND_GAUSS || KIND == LERF_KIND_LINEAR) || !s3::near_tie(res)) return false;
    if (S == 2) *dst = warp_resolve_u8<KIND, 2>(H, W, g, lr, lc, gr, gc, max_sigma, tap);
    else if (S == 4) *dst = warp_resolve_u8<KIND, 4>(H, W, g, lr, lc, gr, gc, max_sigma, tap);
    else return false;
    return true;
}

struct WarpPx2 {
    int lr, lc;                // first tap in PADDED coordinates
    double gr, gc;             // projected position in padded coordinates
    float dx[2], dy[2];        // distances to the two rows / columns of taps
    int cx[2], cy[2];          // their classes for the amplified-linear kernel
    int rrow[2], rcol[2];      // the taps' source rows / columns, clamped into the frame (where the hyper-parameters are read)
    bool in_r[2], in_c[2];     // the tap lies inside the frame (the image is zero outside)
};

__device__ __forceinline__ WarpPx2 warp_px_geometry(const WarpGeo& g, int i, int j, int H, int W) {
    constexpr int S = 2;
    WarpPx2 G;
    double gr, gc;
    project_point(g.minv, i + g.oy0, j + g.ox0, H, W, &gr, &gc);
    G.lr = left_boundary(gr, S) + g.pad_r_lo;
    G.lc = left_boundary(gc, S) + g.pad_c_lo;
    gr += (double)g.pad_r_lo;
    gc += (double)g.pad_c_lo;
    G.gr = gr;
    G.gc = gc;
#pragma unroll
    for (int b = 0; b < S; ++b) {
        const int pr = clampi(G.lr + b, 0, H - 1);
        const double d = gr - (double)pr;
        G.dx[b] = (float)d;
        G.cx[b] = dist_class(d);
        const int sr = pr - g.pad_r_lo;
        G.rrow[b] = clampi(sr, 0, H - 1);
        G.in_r[b] = sr == G.rrow[b];
    }
#pragma unroll
    for (int a = 0; a < S; ++a) {
        const int pc = clampi(G.lc + a, 0, W - 1);
        const double d = gc - (double)pc;
        G.dy[a] = (float)d;
        G.cy[a] = dist_class(d);
        const int sc_ = pc - g.pad_c_lo;
        G.rcol[a] = clampi(sc_, 0, W - 1);
        G.in_c[a] = sc_ == G.rcol[a];
    }
    return G;
}

// channel value of the pixel in production arithmetic; tap(r, c) -> packed dword of the clamped source pixel (this channel).
// Returns true when the byte was written by the tie guard (float64), else *res holds the float32 value to be stored.
template <int KIND, typename Tap>
__device__ __forceinline__ bool warp_px_value_u8(const WarpPx2& G, const WarpGeo& g, int H, int W, float max_sigma, const float (&dxs)[2],
                                                 const float (&dys)[2], Tap tap, uint8_t* dst, float* res_out) {
    constexpr int S = 2;
    const float ms255 = max_sigma * (1.0f / 255.0f);
    uint32_t d[S * S];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) d[a * S + b] = tap(G.rrow[b], G.rcol[a]);
    float e[S * S], v[S * S];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) {
            const uint32_t q = d[a * S + b];
            if (KIND == LERF_KIND_GAUSS) {
                e[a * S + b] = s3::gauss_form_u8((float)(q & 0xFFu), (float)((q >> 8) & 0xFFu), (float)((q >> 16) & 0xFFu), dxs[b], dys[a]);
            } else {
                const float alpha = s3::lin_alpha_u8((float)(q & 0xFFu), ms255);
                e[a * S + b] = s3::lin_factor(alpha, G.dx[b], G.cx[b]) * s3::lin_factor(alpha, G.dy[a], G.cy[a]);
            }
            v[a * S + b] = (G.in_r[b] && G.in_c[a]) ? (float)(q >> 24) : 0.0f;
        }
    float res = s3::finish<KIND == LERF_KIND_GAUSS, S * S, true, true, false>(e, v);
    if (KIND == LERF_KIND_GAUSS) {
        // every weight underflows in the reference's float64 (exp(-e/2) = 0 for e/2 > 745.2): its 0/0 = NaN; in the
        // pre-scaled units e' = 0.5 log2(e) e that is e' > 1075.1
        const float emin = fminf(fminf(e[0], e[1]), fminf(e[2], e[3]));
        if (emin > 1075.1f) res = __builtin_nanf("");
    }
    *res_out = res;
    return warp_tie_guard<KIND>(res, S, H, W, g, G.lr, G.lc, G.gr, G.gc, max_sigma, tap, dst);
}

}  // namespace lerf
