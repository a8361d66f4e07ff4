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

// Backward.  One workgroup owns a band of rows of one plane (about 4096 pixels).  Photo-like patches hit very few LUT
// rows (every pixel of a smooth region lands in the same 16-corner cell), and float atomics to one address serialise at
// the memory side of the 8 XCDs, so the LUT-row contributions are first summed in an LDS table keyed by the row
// (open slot claimed with an LDS compare-and-swap, ds_add_f32 into it; a key that finds its slot taken by another row
// goes to global memory directly) and flushed with one global atomic per occupied entry.  Worst case (uniform noise:
// more distinct rows than slots) costs what the direct version cost; a smooth batch needs ~20x fewer global atomics.
constexpr int BWD_NT = 1024;     // 16 waves per workgroup: the per-pixel chain (image reads -> LUT reads -> LDS CAS) is latency-bound
template <int OC, int LOG_TS>
__global__ void __launch_bounds__(1024)
interp_bwd_kernel(const float* __restrict__ weight, const float* __restrict__ img, const float* __restrict__ gout, int n_planes,
                  int h, int w, int bd, int rows_per_band, int bands, int gim_lds, Pattern pt, float* __restrict__ gweight,
                  float* __restrict__ gimg) {
    constexpr int TS = 1 << LOG_TS;
    __shared__ int tag[TS];
    __shared__ float val[TS * OC];
    extern __shared__ float gim[];          // image-gradient rows of the band (+ bd rows of pattern reach), when it fits
    const int tid = threadIdx.x;
    const int p = blockIdx.x / bands, band = blockIdx.x - p * bands;
    const int y0 = band * rows_per_band, y1 = min(h, y0 + rows_per_band);
    const int wp = w + bd;
    const int gim_n = gim_lds ? (y1 - y0 + bd) * wp : 0;
    for (int i = tid; i < TS; i += BWD_NT) tag[i] = -1;
    for (int i = tid; i < TS * OC; i += BWD_NT) val[i] = 0.0f;
    for (int i = tid; i < gim_n; i += BWD_NT) gim[i] = 0.0f;
    __syncthreads();
    const int64_t poff = (int64_t)p * (h + bd) * wp;
    for (int i = tid; i < (y1 - y0) * w; i += BWD_NT) {
        const int y = y0 + i / w, x = i % w;
        const Walk r = walk_at(img + poff, wp, y, x, pt);
        float g[OC], gf[4] = {0.0f, 0.0f, 0.0f, 0.0f};       // gf: d loss / d f at sorted position n
#pragma unroll
        for (int oc = 0; oc < OC; ++oc) g[oc] = gout[(((int64_t)p * OC + oc) * h + y) * w + x] / (float)kQ;
        float Pprev[OC];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int row = r.idx[n];
            int slot = -1;
            if (gweight && r.w[n] != 0) {
                const int s = (int)(((unsigned)row * 2654435761u) >> (32 - LOG_TS));
                const int old = atomicCAS(&tag[s], -1, row);
                slot = (old == -1 || old == row) ? s : -2;      // -2: table slot belongs to another row
            }
#pragma unroll
            for (int oc = 0; oc < OC; ++oc) {
                const float rq = rintf(weight[(int64_t)row * OC + oc] * 127.0f);
                const float Pn = fminf(fmaxf(rq, -127.0f), 127.0f);
                if (slot != -1 && rq >= -127.0f && rq <= 127.0f) {
                    const float c = g[oc] * (float)r.w[n] * 127.0f;
                    if (slot >= 0) atomicAdd(&val[slot * OC + oc], c);
                    else atomicAdd(gweight + (int64_t)row * OC + oc, c);
                }
                if (n > 0) gf[n - 1] += g[oc] * (Pn - Pprev[oc]);
                Pprev[oc] = Pn;
            }
        }
        if (gimg) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int a = r.axis[n];
                if (gim_lds) atomicAdd(&gim[(y - y0 + pt.ldy[a]) * wp + x + pt.ldx[a]], gf[n]);
                else atomicAdd(gimg + poff + (y + pt.ldy[a]) * wp + x + pt.ldx[a], gf[n]);
            }
        }
    }
    __syncthreads();
    if (gimg && gim_lds) {
        // the band's rows of the plane: exclusive when the plane is one band, shared halo rows otherwise
        float* dst = gimg + poff + (int64_t)y0 * wp;
        for (int i = tid; i < gim_n; i += BWD_NT) {
            const float v = gim[i];
            if (v != 0.0f) {
                if (bands == 1) dst[i] += v;
                else atomicAdd(dst + i, v);
            }
        }
    }
    if (gweight)
        for (int s = tid; s < TS; s += BWD_NT) {
            const int row = tag[s];
            if (row >= 0) {
#pragma unroll
                for (int oc = 0; oc < OC; ++oc) {
                    const float v = val[s * OC + oc];
                    if (v != 0.0f) atomicAdd(gweight + (int64_t)row * OC + oc, v);
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the SR resamplers (SteeringGaussianResize2dTorch.resize / AmplifiedLinearResize2dTorch.resize,
// resize_right2d_torch.py:154-247) on planar float32 maps, as autograd derives it:
//   out = sum_t w_t v_t / W,  W = sum_t w_t          v_t = input at the tap (0 outside the frame)
//   d out / d v_t = w_t / W                           (scattered to the tap's input pixel when inside)
//   d out / d w_t = (v_t - out) / W
//   gauss:  w = exp(-e/2), e = tx^2 - 2 rho tx ty + ty^2, tx = sx dx, ty = sy dy, rho = 2 h0 - 1, s = max_sigma h
//   linear: w = max(lx, 0) max(ly, 0), l(x) = (alpha x + 1)[-1 <= x < 0] + (1 - alpha x)[0 <= x <= 1], alpha = max_sigma (2 h - 1)
// hyper-parameter gradients land on the CLAMPED tap position (replicate padding).  One thread per output element,
// float atomics into the gradient maps (accumulating: the caller zeroes them or passes running sums).
// ---------------------------------------------------------------------------------------------------------------
struct Tap {
    float v, w, dx, dy, tx, ty, rho, alpha, lx, ly, cx, cy;
    int64_t pos;
    bool inside;
};

template <int KIND>
__device__ __forceinline__ Tap load_tap(const float* __restrict__ feat, const float* __restrict__ h0, const float* __restrict__ h1,
                                        const float* __restrict__ h2, int64_t plane, int H, int W, int rr, int cc, float dx, float dy,
                                        float max_sigma) {
    Tap t;
    const int rcl = clampi(rr, 0, H - 1), ccl = clampi(cc, 0, W - 1);
    t.inside = (rr == rcl) && (cc == ccl);
    t.pos = plane + (int64_t)rcl * W + ccl;
    t.v = t.inside ? feat[t.pos] : 0.0f;
    t.dx = dx;
    t.dy = dy;
    if (KIND == LERF_KIND_GAUSS) {
        t.rho = h0[t.pos] * 2.0f - 1.0f;
        t.tx = h1[t.pos] * max_sigma * dx;
        t.ty = h2[t.pos] * max_sigma * dy;
        t.w = t.tx * t.tx - 2.0f * t.rho * (t.tx * t.ty) + t.ty * t.ty;      // the exponent e; turned into a weight by the caller
    } else {
        t.alpha = max_sigma * (h0[t.pos] * 2.0f - 1.0f);
        t.lx = (t.alpha * dx + 1.0f) * ((-1.0f <= dx && dx < 0.0f) ? 1.0f : 0.0f) + (1.0f - t.alpha * dx) * ((0.0f <= dx && dx <= 1.0f) ? 1.0f : 0.0f);
        t.ly = (t.alpha * dy + 1.0f) * ((-1.0f <= dy && dy < 0.0f) ? 1.0f : 0.0f) + (1.0f - t.alpha * dy) * ((0.0f <= dy && dy <= 1.0f) ? 1.0f : 0.0f);
        t.cx = fmaxf(t.lx, 0.0f);
        t.cy = fmaxf(t.ly, 0.0f);
        t.w = t.cx * t.cy;
    }
    return t;
}

// One workgroup = a 16 x 64 block of outputs of one plane.  Its taps fall into a small window of the input (about
// 16/s + S rows by 64/s + S columns), so the four gradient maps of that window are accumulated in LDS (ds_add_f32) and
// flushed with one global atomic per window element: at x4 that is ~40x fewer global atomics than one per tap and map.
// Windows that do not fit (down-sampling) fall back to global atomics per tap.
constexpr int RB_ROWS = 16, RB_COLS = 64, RB_WIN_MAX = (RB_ROWS + LERF_MAX_SUPPORT + 1) * (RB_COLS + LERF_MAX_SUPPORT + 1);

template <int KIND>
__global__ void __launch_bounds__(256)
resize_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ h0, const float* __restrict__ h1,
                  const float* __restrict__ h2, int N, int H, int W, int S, int oH, int oW, const int* __restrict__ left_r,
                  const float* __restrict__ dis_r, const int* __restrict__ left_c, const float* __restrict__ dis_c, float max_sigma,
                  const float* __restrict__ gout, float* __restrict__ gfeat, float* __restrict__ gh0, float* __restrict__ gh1,
                  float* __restrict__ gh2) {
    __shared__ float win[4][RB_WIN_MAX];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * RB_COLS, i0 = blockIdx.y * RB_ROWS, n = blockIdx.z;
    const int i1 = min(i0 + RB_ROWS, oH) - 1, j1 = min(j0 + RB_COLS, oW) - 1;
    // window of clamped source positions touched by this block (left tables are non-decreasing)
    const int wr0 = clampi(left_r[i0], 0, H - 1), wr1 = clampi(left_r[i1] + S - 1, 0, H - 1);
    const int wc0 = clampi(left_c[j0], 0, W - 1), wc1 = clampi(left_c[j1] + S - 1, 0, W - 1);
    const int wh = wr1 - wr0 + 1, ww = wc1 - wc0 + 1;
    const bool lds = wh * ww <= RB_WIN_MAX;
    if (lds) {
        for (int k = tid; k < wh * ww; k += 256) { win[0][k] = 0.0f; win[1][k] = 0.0f; win[2][k] = 0.0f; win[3][k] = 0.0f; }
        __syncthreads();
    }
    const int64_t plane = (int64_t)n * H * W;
    for (int e = tid; e < RB_ROWS * RB_COLS; e += 256) {
        const int i = i0 + e / RB_COLS, j = j0 + e % RB_COLS;
        if (i >= oH || j >= oW) continue;
        const int lr = left_r[i], lc = left_c[j];
        float emin = 0.0f;
        if (KIND == LERF_KIND_GAUSS) {
            for (int a = 0; a < S; ++a)
                for (int b = 0; b < S; ++b) {
                    const Tap t = load_tap<KIND>(feat, h0, h1, h2, plane, H, W, lr + b, lc + a, dis_r[i * S + b], dis_c[j * S + a], max_sigma);
                    emin = (a == 0 && b == 0) ? t.w : fminf(emin, t.w);
                }
        }
        float Wsum = 0.0f, num = 0.0f;
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                Tap t = load_tap<KIND>(feat, h0, h1, h2, plane, H, W, lr + b, lc + a, dis_r[i * S + b], dis_c[j * S + a], max_sigma);
                const float w = KIND == LERF_KIND_GAUSS ? __expf(-0.5f * (t.w - emin)) : t.w;
                Wsum += w;
                num += w * t.v;
            }
        const float out = num / Wsum;
        const float g = gout[((int64_t)n * oH + i) * oW + j];
        for (int a = 0; a < S; ++a)
            for (int b = 0; b < S; ++b) {
                Tap t = load_tap<KIND>(feat, h0, h1, h2, plane, H, W, lr + b, lc + a, dis_r[i * S + b], dis_c[j * S + a], max_sigma);
                const float w = KIND == LERF_KIND_GAUSS ? __expf(-0.5f * (t.w - emin)) : t.w;
                const int64_t rel = t.pos - plane;
                const int k = ((int)(rel / W) - wr0) * ww + ((int)(rel % W) - wc0);        // window slot of the clamped tap
                float gv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (t.inside) gv[0] = g * w / Wsum;
                const float gw = g * (t.v - out) / Wsum;          // d loss / d w_t
                if (KIND == LERF_KIND_GAUSS) {
                    const float c = gw * (-0.5f * w);               // d loss / d e_t
                    gv[1] = c * (-2.0f * t.tx * t.ty) * 2.0f;
                    gv[2] = c * (2.0f * t.dx * (t.tx - t.rho * t.ty)) * max_sigma;
                    gv[3] = c * (2.0f * t.dy * (t.ty - t.rho * t.tx)) * max_sigma;
                } else {
                    // clamp(l, 0) passes the gradient where l >= 0 (torch.clamp backward)
                    const float dlx = (t.lx >= 0.0f ? 1.0f : 0.0f) * (t.dx * ((-1.0f <= t.dx && t.dx < 0.0f) ? 1.0f : 0.0f) - t.dx * ((0.0f <= t.dx && t.dx <= 1.0f) ? 1.0f : 0.0f));
                    const float dly = (t.ly >= 0.0f ? 1.0f : 0.0f) * (t.dy * ((-1.0f <= t.dy && t.dy < 0.0f) ? 1.0f : 0.0f) - t.dy * ((0.0f <= t.dy && t.dy <= 1.0f) ? 1.0f : 0.0f));
                    gv[1] = gw * (dlx * t.cy + t.cx * dly) * 2.0f * max_sigma;
                }
                float* const dst[4] = {gfeat, gh0, gh1, gh2};
#pragma unroll
                for (int m = 0; m < (KIND == LERF_KIND_GAUSS ? 4 : 2); ++m) {
                    if (!dst[m] || (m == 0 && !t.inside)) continue;
                    if (lds) atomicAdd(&win[m][k], gv[m]);
                    else atomicAdd(dst[m] + t.pos, gv[m]);
                }
            }
    }
    if (lds) {
        __syncthreads();
        float* const dst[4] = {gfeat, gh0, gh1, gh2};
        for (int k = tid; k < wh * ww; k += 256) {
            const int64_t pos = plane + (int64_t)(wr0 + k / ww) * W + wc0 + k % ww;
#pragma unroll
            for (int m = 0; m < (KIND == LERF_KIND_GAUSS ? 4 : 2); ++m) {
                const float v = win[m][k];
                if (dst[m] && v != 0.0f) atomicAdd(dst[m] + pos, v);
            }
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
    return lerf::launch_status();
}

int lerf_swf2lut_interp_bwd_f32(const float* weight, int oC, char mode, const float* img, const float* grad_out, int n_planes,
                                int h, int w, int bd, float* grad_weight, float* grad_img, void* stream) {
    Pattern pt;
    if (!weight || !img || !grad_out || n_planes < 1 || h < 1 || w < 1 || bd < 0) return LERF_EINVAL;
    if (!make_pattern(mode, &pt)) return LERF_EINVAL;
    for (int k = 0; k < 4; ++k)
        if (pt.dy[k] > bd || pt.dx[k] > bd || pt.ldy[k] > bd || pt.ldx[k] > bd) return LERF_EINVAL;
    // bands of about 4096 pixels per workgroup
    int rows_per_band = 4096 / w;
    if (rows_per_band < 1) rows_per_band = 1;
    if (rows_per_band > h) rows_per_band = h;
    const int bands = (h + rows_per_band - 1) / rows_per_band;
    dim3 block(BWD_NT), grid((unsigned)(n_planes * bands));
    hipStream_t st = (hipStream_t)stream;
    size_t gim_bytes = grad_img ? (size_t)(rows_per_band + bd) * (size_t)(w + bd) * sizeof(float) : 0;
    const int gim_lds = gim_bytes > 0 && gim_bytes <= 40 * 1024;
    if (!gim_lds) gim_bytes = 0;
    if (oC == 1)
        hipLaunchKernelGGL((interp_bwd_kernel<1, 12>), grid, block, gim_bytes, st, weight, img, grad_out, n_planes, h, w, bd, rows_per_band,
                           bands, gim_lds, pt, grad_weight, grad_img);
    else if (oC == 3)
        hipLaunchKernelGGL((interp_bwd_kernel<3, 12>), grid, block, gim_bytes, st, weight, img, grad_out, n_planes, h, w, bd, rows_per_band,
                           bands, gim_lds, pt, grad_weight, grad_img);
    else
        return LERF_EUNSUPPORTED;
    return lerf::launch_status();
}

int lerf_resize_bwd_f32(const float* feat, const float* h0, const float* h1, const float* h2, int N, int H, int W,
                        const lerf_sr_geo_t* geo, int kind, double max_sigma, const float* grad_out, float* grad_feat,
                        float* grad_h0, float* grad_h1, float* grad_h2, void* stream) {
    if (!feat || !h0 || !geo || !grad_out || N < 1 || H < 1 || W < 1) return LERF_EINVAL;
    if (kind != LERF_KIND_GAUSS && kind != LERF_KIND_LINEAR) return LERF_EUNSUPPORTED;
    if (kind == LERF_KIND_GAUSS && (!h1 || !h2)) return LERF_EINVAL;
    if (!geo->left_r || !geo->left_c || !geo->dis_r || !geo->dis_c || geo->out_h < 1 || geo->out_w < 1) return LERF_EINVAL;
    if (geo->S < 1 || geo->S > LERF_MAX_SUPPORT) return LERF_EUNSUPPORTED;
    dim3 block(256), grid((geo->out_w + RB_COLS - 1) / RB_COLS, (geo->out_h + RB_ROWS - 1) / RB_ROWS, N);
    hipStream_t st = (hipStream_t)stream;
    if (kind == LERF_KIND_GAUSS)
        hipLaunchKernelGGL(resize_bwd_kernel<LERF_KIND_GAUSS>, grid, block, 0, st, feat, h0, h1, h2, N, H, W, geo->S, geo->out_h,
                           geo->out_w, geo->left_r, geo->dis_r, geo->left_c, geo->dis_c, (float)max_sigma, grad_out, grad_feat,
                           grad_h0, grad_h1, grad_h2);
    else
        hipLaunchKernelGGL(resize_bwd_kernel<LERF_KIND_LINEAR>, grid, block, 0, st, feat, h0, h1, h2, N, H, W, geo->S, geo->out_h,
                           geo->out_w, geo->left_r, geo->dis_r, geo->left_c, geo->dis_c, (float)max_sigma, grad_out, grad_feat,
                           grad_h0, nullptr, nullptr);
    return lerf::launch_status();
}

}  // extern "C"
