// Host-side geometry of the LeRF path, free of any HIP dependency: what lerf_api.hip exports as lerf_sr_axis_tables(_f32),
// lerf_out_size, lerf_invert3x3, lerf_warp_pads, lerf_mode_offsets, and the helpers the kernels share with it (pattern
// offsets, homography projection, support boundary).  One source for both builds: liblerf_hip.so (hipcc) and the
// AddressSanitizer / UBSan host build of the same functions (csrc/lerf_host_sanitize.cpp, `make asan`), which the CPU suite
// runs against the product library (tests/test_sanitizers_cpu.py).
#pragma once

#include <math.h>
#include <stdint.h>

#include "lerf_hip.h"

#ifdef __HIPCC__
#define LERF_HD __host__ __device__
#else
#define LERF_HD
#endif

namespace lerf {

constexpr float kEps32 = 1.1920928955078125e-07f;  // np.finfo(np.float32).eps

LERF_HD inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Sampling patterns of resample/eval_lut_sr.py:30-81, (dy, dx) of pixels a,b,c,d.
LERF_HD inline bool mode_pattern(char mode, int8_t dy[4], int8_t dx[4]) {
    dy[0] = 0; dx[0] = 0;
    switch (mode) {
        case 's': dy[1] = 0; dx[1] = 1; dy[2] = 1; dx[2] = 0; dy[3] = 1; dx[3] = 1; return true;
        case 'd': dy[1] = 0; dx[1] = 2; dy[2] = 2; dx[2] = 0; dy[3] = 2; dx[3] = 2; return true;
        case 'y': dy[1] = 1; dx[1] = 1; dy[2] = 1; dx[2] = 2; dy[3] = 2; dx[3] = 1; return true;
        case 'c': dy[1] = 0; dx[1] = 1; dy[2] = 0; dx[2] = 2; dy[3] = 0; dx[3] = 3; return true;
        case 't': dy[1] = 1; dx[1] = 1; dy[2] = 2; dx[2] = 2; dy[3] = 3; dx[3] = 3; return true;
        default: return false;
    }
}

// np.rot90(img, r) + bottom/right edge pad + pattern + rot90 back
// == offsets rotated r times by (dy,dx)->(dx,-dy) with clamped coordinates.
LERF_HD inline bool mode_offsets(char mode, int rot, int8_t dy[4], int8_t dx[4]) {
    if (!mode_pattern(mode, dy, dx)) return false;
    rot &= 3;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < rot; ++r) {
            int8_t t = dy[k];
            dy[k] = dx[k];
            dx[k] = (int8_t)(-t);
        }
    return true;
}

// Homography projection of output pixel (row i, col j) in float64, operation
// order of resize_right/resize_right2d_numpy.py:321-339 (no FMA contraction).
LERF_HD inline void project_point(const double* m, int i, int j, int H, int W, double* gr, double* gc) {
#pragma clang fp contract(off)
    double x = (double)j, y = (double)i;
    double X = m[0] * x + m[1] * y + m[2];
    double Y = m[3] * x + m[4] * y + m[5];
    double Wh = m[6] * x + m[7] * y + m[8];
    X = X / Wh;
    Y = Y / Wh;
    double r = Y, c = X;
    r = r < 0.0 ? 0.0 : (r > (double)H ? (double)H : r);
    c = c < 0.0 ? 0.0 : (c > (double)W ? (double)W : c);
    *gr = r;
    *gc = c;
}

LERF_HD inline int left_boundary(double g, int S) {
#pragma clang fp contract(off)
    return (int)ceil(g - (double)S / 2 - (double)kEps32);
}


namespace host {

// float32 rounding of a float64 distance that keeps its class for the
// amplified-linear kernel's hard masks (resize_right2d_numpy.py:233-235)
inline int dclass(double x) { return (x >= -1.0 && x < 0.0) ? 1 : ((x >= 0.0 && x <= 1.0) ? 2 : 0); }
inline int fclass(float x) { return (x >= -1.0f && x < 0.0f) ? 1 : ((x >= 0.0f && x <= 1.0f) ? 2 : 0); }

inline float class_preserving_f32(double d) {
    float f = (float)d;
    int want = dclass(d);
    if (fclass(f) == want) return f;
    float up = nextafterf(f, INFINITY), dn = nextafterf(f, -INFINITY);
    if (fclass(up) == want) return up;
    if (fclass(dn) == want) return dn;
    return f;
}


inline int out_size(int n_in, double scale) { return (int)ceil(scale * (double)n_in); }

inline int sr_axis_tables(int n_in, int n_out, double scale, int S, int32_t* left, double* dis64, float* dis32,
                        int32_t* pads) {
#pragma clang fp contract(off)
    if (n_in < 1 || n_out < 1 || !(scale > 0.0) || S < 1 || S > LERF_MAX_SUPPORT || !left || !dis64) return LERF_EINVAL;
    // g = i/s + (n_in-1)/2 - (n_out-1)/(2s)            resize_right2d_numpy.py:70-79
    const double a = (double)(n_in - 1) / 2;
    const double b = (double)(n_out - 1) / (2 * scale);
    int pad_lo = 0;
    for (int i = 0; i < n_out; ++i) {
        double g = (double)i / scale + a - b;
        int l = left_boundary(g, S);                    // :85-90
        if (i == 0) pad_lo = -l;                        // :101
        left[i] = l;
        double gp = g + (double)pad_lo;                 // :103
        for (int k = 0; k < S; ++k) {
            double d = gp - (double)(l + pad_lo + k);   // :131-134
            dis64[i * S + k] = d;
            if (dis32) dis32[i * S + k] = class_preserving_f32(d);
        }
    }
    if (pads) {
        pads[0] = pad_lo;
        pads[1] = left[n_out - 1] + S - 1 - n_in + 1;   // :101
    }
    return LERF_OK;
}

inline int sr_axis_tables_f32(int n_in, int n_out, double scale, int S, int32_t* left, float* dis32, int32_t* pads) {
#pragma clang fp contract(off)
    if (n_in < 1 || n_out < 1 || !(scale > 0.0) || S < 1 || S > LERF_MAX_SUPPORT || !left || !dis32) return LERF_EINVAL;
    // Resize2dTorch.get_projected_grid2d / get_field_of_view2d / cal_pad_sz / get_distance
    // (resize_right/resize_right2d_torch.py:48-103): every tensor op is float32, the python scalars are float64
    // expressions rounded to float32 when they meet the tensor.
    const float sf = (float)scale;
    const float a = (float)((double)(n_in - 1) / 2);
    const float b = (float)((double)(n_out - 1) / (2 * scale));
    const float half = (float)((double)S / 2);
    int pad_lo = 0;
    for (int i = 0; i < n_out; ++i) {
        float g = (float)i / sf;                        // :60
        g = g + a;
        g = g - b;
        float t = g - half;                             // :69
        t = t - kEps32;
        const int l = (int)ceilf(t);
        if (i == 0) pad_lo = -l;                        // :81
        left[i] = l;
        const float gp = g + (float)pad_lo;             // :83
        for (int k = 0; k < S; ++k) dis32[i * S + k] = gp - (float)(l + pad_lo + k);      // :98
    }
    if (pads) {
        pads[0] = pad_lo;
        pads[1] = left[n_out - 1] + S - 1 - n_in + 1;   // :81
    }
    return LERF_OK;
}

inline int invert3x3(const double m[9], double out[9]) {
    if (!m || !out) return LERF_EINVAL;
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0) return LERF_EINVAL;
    out[0] = c00 / det;
    out[1] = (m[2] * m[7] - m[1] * m[8]) / det;
    out[2] = (m[1] * m[5] - m[2] * m[4]) / det;
    out[3] = c01 / det;
    out[4] = (m[0] * m[8] - m[2] * m[6]) / det;
    out[5] = (m[2] * m[3] - m[0] * m[5]) / det;
    out[6] = c02 / det;
    out[7] = (m[1] * m[6] - m[0] * m[7]) / det;
    out[8] = (m[0] * m[4] - m[1] * m[3]) / det;
    return LERF_OK;
}

inline int warp_pads(const double minv[9], int in_h, int in_w, int out_h, int out_w, int S, int32_t pads[4]) {
    if (!minv || !pads || in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || S < 1) return LERF_EINVAL;
    double gr, gc;
    project_point(minv, 0, 0, in_h, in_w, &gr, &gc);
    int l0r = left_boundary(gr, S), l0c = left_boundary(gc, S);
    project_point(minv, out_h - 1, out_w - 1, in_h, in_w, &gr, &gc);
    int l1r = left_boundary(gr, S), l1c = left_boundary(gc, S);
    // calc_pad_sz: (max(-fov[0,0],0), max(fov[-1,-1]-in+1,0)), fov[-1,-1] = left + S-1   (:363-366)
    pads[0] = -l0r > 0 ? -l0r : 0;
    pads[1] = (l1r + S - 1 - in_h + 1) > 0 ? (l1r + S - 1 - in_h + 1) : 0;
    pads[2] = -l0c > 0 ? -l0c : 0;
    pads[3] = (l1c + S - 1 - in_w + 1) > 0 ? (l1c + S - 1 - in_w + 1) : 0;
    return LERF_OK;
}

// Tile-fused warp (lerf_warp_fused_u8): a source tile of 64 x 64 pixels OWNS the output pixels whose support (2 x 2 taps,
// clamped into the frame like the reference clips its field of view, resize_right2d_numpy.py:396-398) has its LAST tap row /
// column inside the tile: key = min(first tap + 1, n - 1).  Both taps then lie in the tile or in the ring of one pixel above /
// left of it, which the tile's stage-2 region holds.  (Pixels projected outside the frame are clipped onto its border and belong
// to the border tiles.)
LERF_HD inline void warp_owner_key(const double minv[9], int pad_r_lo, int pad_c_lo, int i, int j, int H, int W, int* key_r, int* key_c) {
    double gr, gc;
    project_point(minv, i, j, H, W, &gr, &gc);
    const int lr = left_boundary(gr, 2) + pad_r_lo, lc = left_boundary(gc, 2) + pad_c_lo;
    const int r0 = clampi(clampi(lr, 0, H - 1) - pad_r_lo, 0, H - 1), c0 = clampi(clampi(lc, 0, W - 1) - pad_c_lo, 0, W - 1);
    *key_r = r0 + 1 < H - 1 ? r0 + 1 : H - 1;
    *key_c = c0 + 1 < W - 1 ? c0 + 1 : W - 1;
}

// boxes[t] = {i0, i1, j0, j1}: output rows [i0, i1) x columns [j0, j1) that contain every pixel tile t (row-major tiles of
// `tile` x `tile` source pixels) owns, widened by 2 pixels (the kernel decides ownership itself, pixel by pixel, with the
// same formula; the boxes only bound its search).  One pass over the output on the host, once per homography.
inline int warp_tile_boxes(const double minv[9], int pad_r_lo, int pad_c_lo, int H, int W, int oH, int oW, int tile, int32_t* boxes) {
    if (!minv || !boxes || H < 1 || W < 1 || oH < 1 || oW < 1 || tile < 1) return LERF_EINVAL;
    const int ty = (H + tile - 1) / tile, tx = (W + tile - 1) / tile;
    for (int t = 0; t < ty * tx; ++t) { boxes[4 * t] = oH; boxes[4 * t + 1] = -1; boxes[4 * t + 2] = oW; boxes[4 * t + 3] = -1; }
    for (int i = 0; i < oH; ++i)
        for (int j = 0; j < oW; ++j) {
            int kr, kc;
            warp_owner_key(minv, pad_r_lo, pad_c_lo, i, j, H, W, &kr, &kc);
            int32_t* b = boxes + 4 * ((kr / tile) * tx + kc / tile);
            if (i < b[0]) b[0] = i;
            if (i > b[1]) b[1] = i;
            if (j < b[2]) b[2] = j;
            if (j > b[3]) b[3] = j;
        }
    for (int t = 0; t < ty * tx; ++t) {
        int32_t* b = boxes + 4 * t;
        if (b[1] < 0) { b[0] = b[1] = b[2] = b[3] = 0; continue; }
        b[0] = b[0] - 2 > 0 ? b[0] - 2 : 0;
        b[1] = b[1] + 3 < oH ? b[1] + 3 : oH;
        b[2] = b[2] - 2 > 0 ? b[2] - 2 : 0;
        b[3] = b[3] + 3 < oW ? b[3] + 3 : oW;
    }
    return LERF_OK;
}

}  // namespace host
}  // namespace lerf
