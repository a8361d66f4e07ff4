/*
 * lerf_oracle.c -- plain-C CPU restatement of the LeRF LUT resampling path.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path at sizes where the
 * numpy oracle (lerf_oracle.py) is too slow, and the "port" CPU baseline that
 * bench.py times beside the GPU.  Nothing in the product package links or
 * loads this file.  Parity: pinned -- tests/test_oracle_c.py checks it against
 * the reference-generated golden vectors and the numpy oracle.
 *
 * Reference citations (paths relative to the upstream repository):
 *   simplex interpolation ........ resample/eval_lut_sr.py:24-470
 *   stage ensembles, rounding .... resample/eval_lut_sr.py:541-628
 *   SR geometry .................. resize_right/resize_right2d_numpy.py:57-140
 *   steering Gaussian / linear ... resize_right/resize_right2d_numpy.py:142-282
 *   final rounding ............... resample/eval_lut_sr.py:663-665
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define L1 17
#define Q 16
static const int STRIDE[4] = {L1 * L1 * L1, L1 * L1, L1, 1};

/* eval_lut_sr.py:30-81 */
static int pattern(char mode, int dy[4], int dx[4]) {
    static const int S_[8] = {0, 0, 0, 1, 1, 0, 1, 1}, D_[8] = {0, 0, 0, 2, 2, 0, 2, 2}, Y_[8] = {0, 0, 1, 1, 1, 2, 2, 1},
                     C_[8] = {0, 0, 0, 1, 0, 2, 0, 3}, T_[8] = {0, 0, 1, 1, 2, 2, 3, 3};
    const int* p;
    switch (mode) {
        case 's': p = S_; break;
        case 'd': p = D_; break;
        case 'y': p = Y_; break;
        case 'c': p = C_; break;
        case 't': p = T_; break;
        default: return -1;
    }
    for (int k = 0; k < 4; ++k) { dy[k] = p[2 * k]; dx[k] = p[2 * k + 1]; }
    return 0;
}

/* np.rot90(img, r) + edge pad + rot90 back == rotate offsets, clamp coordinates (eval_lut_sr.py:549-553, 468) */
static int rotated(char mode, int r, int dy[4], int dx[4]) {
    if (pattern(mode, dy, dx)) return -1;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < (r & 3); ++i) { int t = dy[k]; dy[k] = dx[k]; dx[k] = -t; }
    return 0;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* 24-case ordering of eval_lut_sr.py:218-462 as: walk the axes by decreasing LSB */
static inline void simplex(const int8_t* lut, int oC, const int v[4], int* acc) {
    int f[4], ax[4], idx = 0;
    for (int k = 0; k < 4; ++k) { idx += (v[k] >> 4) * STRIDE[k]; f[k] = v[k] & 15; ax[k] = k; }
    for (int i = 1; i < 4; ++i) {                       /* stable insertion sort, descending */
        int fi = f[i], ai = ax[i], j = i - 1;
        while (j >= 0 && f[j] < fi) { f[j + 1] = f[j]; ax[j + 1] = ax[j]; --j; }
        f[j + 1] = fi; ax[j + 1] = ai;
    }
    int w = Q - f[0];
    for (int c = 0; c < oC; ++c) acc[c] += w * lut[(size_t)idx * oC + c];
    for (int n = 0; n < 4; ++n) {
        idx += STRIDE[ax[n]];
        w = f[n] - (n < 3 ? f[n + 1] : 0);
        for (int c = 0; c < oC; ++c) acc[c] += w * lut[(size_t)idx * oC + c];
    }
}

/* np.round (half to even) of n/d, clipped to [0,255] */
static inline int rne_div_clip(int n, int d) {
    if (n <= 0) return 0;
    int q = n / d, r = n - q * d;
    q += (2 * r > d) || (2 * r == d && (q & 1));
    return q > 255 ? 255 : q;
}

/* one LUT stage over an HWC uint8 image; luts[m*2+parity]; out [H][W][C][oC] */
static int lut_stage(const uint8_t* img, int H, int W, int C, const char* modes, int n_modes, const int8_t* const* luts,
                     int oC, int div, int bias, uint8_t* out) {
    int dy[5][4][4], dx[5][4][4];
    if (n_modes < 1 || n_modes > 5) return -1;
    for (int m = 0; m < n_modes; ++m)
        for (int r = 0; r < 4; ++r)
            if (rotated(modes[m], r, dy[m][r], dx[m][r])) return -1;
    /* rows are dealt out dynamically in small chunks: with one static slab per thread the slowest hardware thread (SMT
     * siblings, a busy core) sets the time of the whole stage */
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                int acc[3] = {0, 0, 0};
                for (int m = 0; m < n_modes; ++m)
                    for (int r = 0; r < 4; ++r) {
                        int v[4];
                        for (int k = 0; k < 4; ++k) {
                            int yy = clampi(y + dy[m][r][k], 0, H - 1), xx = clampi(x + dx[m][r][k], 0, W - 1);
                            v[k] = img[((size_t)yy * W + xx) * C + c];
                        }
                        simplex(luts[m * 2 + (r & 1)], oC, v, acc);
                    }
                for (int k = 0; k < oC; ++k)
                    out[(((size_t)y * W + x) * C + c) * oC + k] = (uint8_t)rne_div_clip(acc[k] + bias * div, div);
            }
    return 0;
}

/* stages 1+2 (eval_lut_sr.py:541-628).  s1[m]: int8 [17^4]; s2[m*2+p]: int8 [17^4][oC] */
int lerf_oracle_lut_stages(const uint8_t* img, int H, int W, int C, const char* modes1, int n1, const int8_t* const* s1,
                           const char* modes2, int n2, const int8_t* const* s2, int oC, uint8_t* feat, uint8_t* hq) {
    const int8_t* l1[10];
    for (int m = 0; m < n1; ++m) l1[2 * m] = l1[2 * m + 1] = s1[m];
    if (lut_stage(img, H, W, C, modes1, n1, l1, 1, Q * n1, 0, feat)) return -1;
    if (!hq) return 0;
    return lut_stage(feat, H, W, C, modes2, n2, s2, oC, Q * 4 * n2, 127, hq);
}

/* 1-D tables of Resize2dNumpy (resize_right2d_numpy.py:70-79, 85-90, 100-104, 131-134) */
static void axis_tables(int n_in, int n_out, double s, int S, int* left, double* dis) {
    const double eps = 1.1920928955078125e-07;
    const double a = (double)(n_in - 1) / 2, b = (double)(n_out - 1) / (2 * s);
    int pad = 0;
    for (int i = 0; i < n_out; ++i) {
        double g = (double)i / s + a - b;
        int l = (int)ceil(g - (double)S / 2 - eps);
        if (i == 0) pad = -l;
        left[i] = l;
        double gp = g + (double)pad;
        for (int k = 0; k < S; ++k) dis[i * S + k] = gp - (double)(l + pad + k);
    }
}

static inline double lin_alpha(double x, double a) {   /* resize_right2d_numpy.py:233-235 */
    double r = 0;
    if (-1 <= x && x < 0) r += a * x + 1;
    if (0 <= x && x <= 1) r += 1 - a * x;
    return r;
}

/* stage 3 on the uint8 stage outputs (resize_right2d_numpy.py:162-223, 243-282).
 * feat [H][W][C], hq [H][W][C][oC]; out float64 [oH][oW][C] and / or out8 uint8 [oH][oW][C] = clip(np.round(.)) of it
 * (eval_lut_sr.py:663-665; either may be NULL); kind 0 = gauss, 1 = linear.  tables: scratch for the 1-D geometry,
 * (oH + oW) ints + (oH + oW) * S doubles, or NULL (allocated here). */
static int resize_core(const uint8_t* feat, const uint8_t* hq, int H, int W, int C, int oC, double sh, double sw, int S,
                       double max_sigma, int kind, double* out, uint8_t* out8, void* tables) {
    int oH = (int)ceil(sh * H), oW = (int)ceil(sw * W);
    void* own = NULL;
    if (!tables) {
        own = malloc(sizeof(double) * (size_t)(oH + oW) * S + sizeof(int) * (size_t)(oH + oW));
        if (!own) return -1;
        tables = own;
    }
    double* dr = (double*)tables;
    double* dc = dr + (size_t)oH * S;
    int* lr = (int*)(dc + (size_t)oW * S);
    int* lc = lr + oH;
    axis_tables(H, oH, sh, S, lr, dr);
    axis_tables(W, oW, sw, S, lc, dc);
    const float ms = (float)max_sigma;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < oH; ++i)
        for (int j = 0; j < oW; ++j)
            for (int c = 0; c < C; ++c) {
                double num = 0, den = 0;
                for (int a = 0; a < S; ++a)          /* column offset major (numpy meshgrid 'xy', :95-98) */
                    for (int b = 0; b < S; ++b) {
                        int rr = lr[i] + b, cc = lc[j] + a;
                        int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);
                        double val = (rr == rcl && cc == ccl) ? (double)feat[((size_t)rcl * W + ccl) * C + c] : 0.0;
                        const uint8_t* h = hq + (((size_t)rcl * W + ccl) * C + c) * oC;
                        double dx = dr[i * S + b], dy = dc[j * S + a], w;
                        if (kind == 0) {
                            /* float32 parameter formation (:168-170), float64 afterwards */
                            float h0 = (float)h[0] / 255.0f, h1 = (float)h[1] / 255.0f, h2 = (float)h[2] / 255.0f;
                            double rho = (double)(h0 * 2.0f - 1.0f), sx = (double)(h1 * ms), sy = (double)(h2 * ms);
                            double xn = (sx * dx) * (sx * dx), yn = (sy * dy) * (sy * dy), xy = sx * dx * sy * dy;
                            w = exp(-0.5 * (xn - 2 * rho * xy + yn));
                        } else {
                            float h0 = (float)h[0] / 255.0f;
                            double al = (double)(ms * (h0 * 2.0f - 1.0f));
                            double wx = lin_alpha(dx, al), wy = lin_alpha(dy, al);
                            w = (wx < 0 ? 0 : wx) * (wy < 0 ? 0 : wy);
                        }
                        num += w * val;
                        den += w;
                    }
                const double v = num / den;
                const size_t k = ((size_t)i * oW + j) * C + c;
                if (out) out[k] = v;
                if (out8) {
                    double r = nearbyint(v);                 /* np.round: half to even (:663-665) */
                    out8[k] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
            }
    free(own);
    return 0;
}

int lerf_oracle_resize(const uint8_t* feat, const uint8_t* hq, int H, int W, int C, int oC, double sh, double sw, int S,
                       double max_sigma, int kind, double* out) {
    return resize_core(feat, hq, H, W, C, oC, sh, sw, S, max_sigma, kind, out, NULL, NULL);
}

/* whole SR path, uint8 HWC -> uint8 HWC (eval_lut_sr.py:541-665).
 * scratch: caller-owned work area of lerf_oracle_sr_scratch_bytes() bytes (feat, hyper numerators, geometry tables), or
 * NULL (allocated and freed per call).  A timing loop passes one buffer for all its calls: malloc'ing a quarter of a
 * gigabyte per frame and first-touching it from 128 threads is page-fault time, not LeRF time.  The float64 frame of the
 * reference is never materialised here: every value is rounded where it is produced (same bytes). */
size_t lerf_oracle_sr_scratch_bytes(int H, int W, int C, int oC, double sh, double sw, int S) {
    int oH = (int)ceil(sh * H), oW = (int)ceil(sw * W);
    size_t a = ((size_t)H * W * C + 63) / 64 * 64, b = ((size_t)H * W * C * oC + 63) / 64 * 64;
    return a + b + sizeof(double) * (size_t)(oH + oW) * S + sizeof(int) * (size_t)(oH + oW) + 64;
}

int lerf_oracle_sr_u8_ws(const uint8_t* img, int H, int W, int C, const char* modes1, int n1, const int8_t* const* s1,
                         const char* modes2, int n2, const int8_t* const* s2, int oC, double sh, double sw, int S,
                         double max_sigma, int kind, uint8_t* out, void* scratch) {
    void* own = NULL;
    if (!scratch) {
        own = malloc(lerf_oracle_sr_scratch_bytes(H, W, C, oC, sh, sw, S));
        if (!own) return -1;
        scratch = own;
    }
    uint8_t* feat = (uint8_t*)scratch;
    uint8_t* hq = feat + ((size_t)H * W * C + 63) / 64 * 64;
    void* tables = hq + ((size_t)H * W * C * oC + 63) / 64 * 64;
    tables = (void*)(((uintptr_t)tables + 7) & ~(uintptr_t)7);
    int rc = lerf_oracle_lut_stages(img, H, W, C, modes1, n1, s1, modes2, n2, s2, oC, feat, hq);
    if (!rc) rc = resize_core(feat, hq, H, W, C, oC, sh, sw, S, max_sigma, kind, NULL, out, tables);
    free(own);
    return rc;
}

int lerf_oracle_sr_u8(const uint8_t* img, int H, int W, int C, const char* modes1, int n1, const int8_t* const* s1,
                      const char* modes2, int n2, const int8_t* const* s2, int oC, double sh, double sw, int S,
                      double max_sigma, int kind, uint8_t* out) {
    return lerf_oracle_sr_u8_ws(img, H, W, C, modes1, n1, s1, modes2, n2, s2, oC, sh, sw, S, max_sigma, kind, out, NULL);
}

/* ---------------------------------------------------------------------------------------------------------
 * homographic warp (resize_right/resize_right2d_numpy.py:284-449 geometry, :496-577 Gaussian, :579-636 linear,
 * :460-467 + interp_methods.py:67-70 nearest/box).  minv: the INVERSE matrix (np.linalg.inv of the caller, :327).
 * ------------------------------------------------------------------------------------------------------- */
static inline void project(const double* m, int i, int j, int H, int W, double* gr, double* gc) {
    /* (x, y, 1) = (col, row, 1); inv(M) . p; perspective divide; flip; clip to [0, in_sz]  (:318-339) */
    double x = (double)j, y = (double)i;
    double X = m[0] * x + m[1] * y + m[2];
    double Y = m[3] * x + m[4] * y + m[5];
    double Wh = m[6] * x + m[7] * y + m[8];
    X = X / Wh;
    Y = Y / Wh;
    *gr = Y < 0.0 ? 0.0 : (Y > (double)H ? (double)H : Y);
    *gc = X < 0.0 ? 0.0 : (X > (double)W ? (double)W : X);
}

static inline int left_of(double g, int S) {
    const double eps = 1.1920928955078125e-07;
    return (int)ceil(g - (double)S / 2 - eps);                         /* :344-350 */
}

/* pads {r_lo, r_hi, c_lo, c_hi} from the two corner pixels only (:363-369) */
int lerf_oracle_warp_pads(const double* minv, int H, int W, int oH, int oW, int S, int* pads) {
    double gr, gc;
    project(minv, 0, 0, H, W, &gr, &gc);
    int lr0 = left_of(gr, S), lc0 = left_of(gc, S);
    project(minv, oH - 1, oW - 1, H, W, &gr, &gc);
    int lr1 = left_of(gr, S), lc1 = left_of(gc, S);
    pads[0] = -lr0 > 0 ? -lr0 : 0;
    pads[1] = lr1 + S - 1 - H + 1 > 0 ? lr1 + S - 1 - H + 1 : 0;
    pads[2] = -lc0 > 0 ? -lc0 : 0;
    pads[3] = lc1 + S - 1 - W + 1 > 0 ? lc1 + S - 1 - W + 1 : 0;
    return 0;
}

static inline double box1(double x) {                                   /* interp_methods.py:67-70 */
    return (double)(-1 <= x && x < 0) + (double)(0 <= x && x <= 1);
}

/* feat uint8 [H][W][C], hq uint8 [H][W][C][oC] (NULL for kind 2); out float64 [oH][oW][C] (NaN where every weight
 * vanishes, the reference's 0/0); kind 0 = gauss, 1 = linear, 2 = nearest (box).
 * The padded arrays are never materialised: index f into the padded frame = source index f - pad_lo, outside of
 * which the image is 0 (constant pad) and the hyper maps take the clamped pixel (edge pad) -- with the
 * reference's quirk that f is clipped to [0, in_sz - 1] in PADDED coordinates (:396-398). */
int lerf_oracle_warp(const uint8_t* feat, const uint8_t* hq, int H, int W, int C, int oC, const double* minv, int oH, int oW,
                     int S, double max_sigma, int kind, double* out) {
    int pads[4];
    lerf_oracle_warp_pads(minv, H, W, oH, oW, S, pads);
    const int plr = pads[0], plc = pads[2];
    const float ms = (float)max_sigma;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < oH; ++i)
        for (int j = 0; j < oW; ++j) {
            double gr, gc;
            project(minv, i, j, H, W, &gr, &gc);
            const int lr = left_of(gr, S) + plr, lc = left_of(gc, S) + plc;
            const double grp = gr + (double)plr, gcp = gc + (double)plc;
            for (int c = 0; c < C; ++c) {
                double num = 0, den = 0;
                for (int a = 0; a < S; ++a)
                    for (int b = 0; b < S; ++b) {
                        const int fr = clampi(lr + b, 0, H - 1), fc = clampi(lc + a, 0, W - 1);   /* padded coords */
                        const double dx = grp - (double)fr, dy = gcp - (double)fc;
                        const int sr = fr - plr, sc = fc - plc;                                     /* source coords */
                        const int rcl = clampi(sr, 0, H - 1), ccl = clampi(sc, 0, W - 1);
                        const double val = (sr == rcl && sc == ccl) ? (double)feat[((size_t)rcl * W + ccl) * C + c] : 0.0;
                        double w;
                        if (kind == 0) {
                            const uint8_t* h = hq + (((size_t)rcl * W + ccl) * C + c) * oC;
                            float h0 = (float)h[0] / 255.0f, h1 = (float)h[1] / 255.0f, h2 = (float)h[2] / 255.0f;
                            double rho = (double)(h0 * 2.0f - 1.0f), sx = (double)(h1 * ms), sy = (double)(h2 * ms);
                            double xn = (sx * dx) * (sx * dx), yn = (sy * dy) * (sy * dy), xy = sx * dx * sy * dy;
                            w = exp(-0.5 * (xn - 2 * rho * xy + yn));
                        } else if (kind == 1) {
                            const uint8_t* h = hq + (((size_t)rcl * W + ccl) * C + c) * oC;
                            float h0 = (float)h[0] / 255.0f;
                            double al = (double)(ms * (h0 * 2.0f - 1.0f));
                            double wx = lin_alpha(dx, al), wy = lin_alpha(dy, al);
                            w = (wx < 0 ? 0 : wx) * (wy < 0 ? 0 : wy);
                        } else {
                            w = box1(dx) * box1(dy);
                        }
                        num += w * val;
                        den += w;
                    }
                out[((size_t)i * oW + j) * C + c] = num / den;            /* 0/0 -> NaN like numpy */
            }
        }
    return 0;
}

/* whole warp path of eltr._worker (eval_lut_warp.py:100-233): uint8 HWC in -> uint8 HWC out (NaN -> 0) and the
 * validity mask (nearest warp of a white frame with a `border`-px black rim, == 255; :197-204, 229).
 * mask may be NULL. */
int lerf_oracle_warp_u8(const uint8_t* img, int H, int W, int C, const char* modes1, int n1, const int8_t* const* s1,
                        const char* modes2, int n2, const int8_t* const* s2, int oC, const double* minv, int oH, int oW,
                        int S, double max_sigma, int kind, int border, uint8_t* out, uint8_t* mask) {
    uint8_t* feat = (uint8_t*)malloc((size_t)H * W * C);
    uint8_t* hq = (uint8_t*)malloc((size_t)H * W * C * oC);
    double* o = (double*)malloc(sizeof(double) * (size_t)oH * oW * C);
    if (!feat || !hq || !o) return -1;
    int rc = lerf_oracle_lut_stages(img, H, W, C, modes1, n1, s1, modes2, n2, s2, oC, feat, hq);
    if (!rc) rc = lerf_oracle_warp(feat, hq, H, W, C, oC, minv, oH, oW, S, max_sigma, kind, o);
    const size_t n = (size_t)oH * oW * C;
    if (!rc) {
#pragma omp parallel for schedule(static)
        for (size_t k = 0; k < n; ++k) {
            double r = nearbyint(o[k]);                      /* NaN compares false everywhere -> 0 */
            out[k] = (uint8_t)(r > 0 ? (r > 255 ? 255 : r) : 0);
        }
    }
    if (!rc && mask) {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < C; ++c)
                    feat[((size_t)y * W + x) * C + c] =
                        (y >= border && y < H - border && x >= border && x < W - border) ? 255 : 0;
        rc = lerf_oracle_warp(feat, NULL, H, W, C, 0, minv, oH, oW, 1, 1.0, 2, o);
        if (!rc) {
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < n; ++k) mask[k] = o[k] == 255.0;
        }
    }
    free(feat); free(hq); free(o);
    return rc;
}

/* n > 0: use n OpenMP threads from now on; n <= 0: back to one per host core */
void lerf_oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}

int lerf_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
