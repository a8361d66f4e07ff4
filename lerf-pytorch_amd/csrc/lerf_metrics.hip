// Evaluation metrics of the reference harness on the device: Y-channel PSNR and SSIM of
// resample/eval_lut_sr.py:741-743 (common/utils.py:46-76 _rgb2ycbcr, :138-151 PSNR, :177-206 cal_ssim) and the masked
// mPSNR of resample/eval_lut_warp.py:233 (common/utils.py:168-175).  Each entry point leaves two doubles in `result`
// (a sum and a count); the host turns them into dB.  uint8 HWC RGB frames, row pitch in elements.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lerf_common.h"

namespace lerf {
namespace metrics {

// Y of ITU-R BT.601 as _rgb2ycbcr forms it: float64 dot product with the first row of T, + 16 (utils.py:54-69)
__device__ __forceinline__ double luma(const uint8_t* p) {
    return 0.256788235294118 * (double)p[0] + 0.504129411764706 * (double)p[1] + 0.097905882352941 * (double)p[2] + 16.0;
}

// sum over the 256-thread block -> thread 0
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// PSNR(y_true, y_pred, shave): both Y planes are cast to float32, the difference and its square are float32 (utils.py:143-149)
__global__ void __launch_bounds__(256) y_sse_kernel(const uint8_t* __restrict__ gt, int64_t gt_sy, const uint8_t* __restrict__ out,
                                                    int64_t out_sy, int h, int w, int shave, double* __restrict__ result) {
    __shared__ double red[4];
    const int64_t n = (int64_t)h * w;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w) + shave, x = (int)(i % w) + shave;
        const float d = (float)luma(out + y * out_sy + 3 * x) - (float)luma(gt + y * gt_sy + 3 * x);
        acc += (double)(d * d);
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) {
        atomicAdd(result, s);
        if (blockIdx.x == 0) result[1] = (double)h * (double)w;
    }
}

// mPSNR: diff = mask * (sr - hr) / 255 in float32, squared in float32 (utils.py:170-174)
__global__ void __launch_bounds__(256) masked_sse_kernel(const uint8_t* __restrict__ sr, const uint8_t* __restrict__ hr,
                                                         const uint8_t* __restrict__ mask, int64_t n, double* __restrict__ result) {
    __shared__ double red[4];
    double acc = 0.0, cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[i] ? 1.0f : 0.0f;
        const float d = m * ((float)sr[i] - (float)hr[i]) / 255.0f;
        acc += (double)(d * d);
        cnt += (double)m;
    }
    const double s = block_sum(acc, red);
    const double c = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        atomicAdd(result, s);
        atomicAdd(result + 1, c);
    }
}

// cal_ssim: 11x11 Gaussian window (sigma 1.5), 'valid' convolution, float64 throughout.  One block = 16x16 outputs;
// the 26x26 Y patches of both frames go through LDS, the window is applied separably (rows, then columns).
constexpr int SW = 11, TO = 16, TI = TO + SW - 1;
struct SsimWin { double k[SW]; };

__global__ void __launch_bounds__(256) ssim_y_kernel(const uint8_t* __restrict__ gt, int64_t gt_sy, const uint8_t* __restrict__ out,
                                                     int64_t out_sy, int H, int W, SsimWin win, double* __restrict__ result) {
    __shared__ double ya[TI][TI + 1], yb[TI][TI + 1];
    __shared__ double hrow[5][TI][TO + 1];
    __shared__ double red[4];
    const int oh = H - SW + 1, ow = W - SW + 1;
    const int by = blockIdx.y * TO, bx = blockIdx.x * TO;
    for (int i = threadIdx.x; i < TI * TI; i += 256) {
        const int r = i / TI, c = i % TI;
        const int y = min(by + r, H - 1), x = min(bx + c, W - 1);
        ya[r][c] = luma(gt + y * gt_sy + 3 * x);
        yb[r][c] = luma(out + y * out_sy + 3 * x);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TI * TO; i += 256) {
        const int r = i / TO, c = i % TO;
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
        for (int t = 0; t < SW; ++t) {
            const double a = ya[r][c + t], b = yb[r][c + t], k = win.k[t];
            m1 += k * a;
            m2 += k * b;
            s11 += k * (a * a);
            s22 += k * (b * b);
            s12 += k * (a * b);
        }
        hrow[0][r][c] = m1; hrow[1][r][c] = m2; hrow[2][r][c] = s11; hrow[3][r][c] = s22; hrow[4][r][c] = s12;
    }
    __syncthreads();
    const int r = threadIdx.x / TO, c = threadIdx.x % TO;
    double v = 0.0;
    if (by + r < oh && bx + c < ow) {
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
#pragma unroll
        for (int t = 0; t < SW; ++t) {
            const double k = win.k[t];
            m1 += k * hrow[0][r + t][c];
            m2 += k * hrow[1][r + t][c];
            s11 += k * hrow[2][r + t][c];
            s22 += k * hrow[3][r + t][c];
            s12 += k * hrow[4][r + t][c];
        }
        const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
        const double m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        v = ((2 * m12 + C1) * (2 * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
    }
    const double s = block_sum(v, red);
    if (threadIdx.x == 0) {
        atomicAdd(result, s);
        if (blockIdx.x == 0 && blockIdx.y == 0) result[1] = (double)oh * (double)ow;
    }
}

static int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace metrics
}  // namespace lerf

using namespace lerf::metrics;

extern "C" {

int lerf_metric_y_sse_u8(const uint8_t* gt, int64_t gt_sy, const uint8_t* out, int64_t out_sy, int H, int W, int shave,
                         double* result, void* stream) {
    if (!gt || !out || !result || H <= 0 || W <= 0 || shave < 0 || H - 2 * shave <= 0 || W - 2 * shave <= 0) return LERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int h = H - 2 * shave, w = W - 2 * shave;
    if (hipMemsetAsync(result, 0, 2 * sizeof(double), st) != hipSuccess) return LERF_ELAUNCH;
    hipLaunchKernelGGL(y_sse_kernel, dim3(grid_for((int64_t)h * w)), dim3(256), 0, st, gt, gt_sy, out, out_sy, h, w, shave, result);
    return lerf::launch_status();
}

int lerf_metric_ssim_y_u8(const uint8_t* gt, int64_t gt_sy, const uint8_t* out, int64_t out_sy, int H, int W,
                          double* result, void* stream) {
    if (!gt || !out || !result || H < SW || W < SW) return LERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // cv2.getGaussianKernel(11, 1.5) in float64: exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1 (utils.py:180)
    SsimWin win;
    double sum = 0.0;
    for (int i = 0; i < SW; ++i) {
        const double x = i - (SW - 1) / 2.0;
        win.k[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
        sum += win.k[i];
    }
    for (int i = 0; i < SW; ++i) win.k[i] /= sum;
    const int oh = H - SW + 1, ow = W - SW + 1;
    if (hipMemsetAsync(result, 0, 2 * sizeof(double), st) != hipSuccess) return LERF_ELAUNCH;
    hipLaunchKernelGGL(ssim_y_kernel, dim3((ow + TO - 1) / TO, (oh + TO - 1) / TO), dim3(256), 0, st, gt, gt_sy, out, out_sy, H, W,
                       win, result);
    return lerf::launch_status();
}

int lerf_metric_masked_sse_u8(const uint8_t* sr, const uint8_t* hr, const uint8_t* mask, int64_t n, double* result, void* stream) {
    if (!sr || !hr || !mask || !result || n <= 0) return LERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(result, 0, 2 * sizeof(double), st) != hipSuccess) return LERF_ELAUNCH;
    hipLaunchKernelGGL(masked_sse_kernel, dim3(grid_for(n)), dim3(256), 0, st, sr, hr, mask, n, result);
    return lerf::launch_status();
}

}  // extern "C"
