// Net -> LUT transfer on MI355X (gfx950): one SRNet hyper-network of the reference evaluated on all L^4 sampled pixel
// tuples and quantised to the int8 LUT the deploy path reads.
//
// Reference being replaced: resample/transfer_to_lut.py:12-42 (get_input_tensor: the L^4 x 4 grid of sampled values,
// first pixel = slowest axis), :45-81 (get_mode_input_tensor: where the four values sit in the net's receptive field
// -- a permutation-free placement, the flattened first-layer kernel meets them in the order a,b,c,d for every mode),
// :96-133 / :136-170 (forward in 100 batches, round(clamp(y,-1,1)*127) -> int8) and the network itself,
// common/network.py:40-71 (SRUnit: conv1 + ReLU, four dense 1x1 layers with concatenation, conv6 + tanh) under
// SRNet.forward's unfold/fold (:127-163), which for a single receptive field is the identity.
//
// This is the one dense contraction of the whole path: per tuple 4->64, 64->64, 128->64, 192->64, 256->64, 320->outC
// (41 k multiply-adds), 83 521 tuples per LUT.  The four hidden layers run on the matrix cores with the float32-input
// MFMA (v_mfma_f32_32x32x2_f32: exact float32 products and sums, no reduced-precision inputs), 64 tuples per workgroup,
// activations resident in LDS ([64][320] floats), weights streamed from L2.  3.4 GFLOP per LUT: microseconds of MFMA
// time; the kernel is written for exactness and clarity, not tuned.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lerf_kernels.h"

namespace lerf {

namespace transfer {
constexpr int ROWS = 64;             // tuples per workgroup
constexpr int NF = 64;               // hidden width (option.py: --nf 64)
constexpr int ACT = 5 * NF;          // 320 concatenated activations
constexpr int PITCH = ACT + 4;       // LDS row pitch in floats (bank spread)
typedef float floatx16 __attribute__((ext_vector_type(16)));

// packed weights of one SRNet (floats): W1[64][4] b1[64] W2[64][64] b2[64] W3[64][128] b3[64] W4[64][192] b4[64]
// W5[64][256] b5[64] W6[outC][320] b6[outC]
__host__ __device__ constexpr int off_w(int layer) {      // layer 1..6 -> offset of W_layer
    int o = 0;
    for (int l = 1; l < layer; ++l) o += NF * (l == 1 ? 4 : (l - 1) * NF) + NF;
    return o;
}

__global__ void __launch_bounds__(256)
srnet_lut_kernel(const float* __restrict__ W, int outC, int interval, int L, int n_entries, int8_t* __restrict__ lut,
                 float* __restrict__ yout) {
    extern __shared__ __attribute__((aligned(16))) float act[];       // [ROWS][PITCH], 82 944 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;

    // ---- inputs + conv1 (4 -> 64) + ReLU: thread = (row, 16 neurons)
    {
        const int r = tid >> 2, n0 = (tid & 3) * 16;
        int e = min(row0 + r, n_entries - 1);
        float x[4];
#pragma unroll
        for (int k = 3; k >= 0; --k) {                     // digit 0 (pixel a) is the slowest axis
            const int dgt = e % L;
            e /= L;
            const int v = min(dgt << interval, 255);       // base = 0, 16, ..., 240, 255  (transfer_to_lut.py:14-15)
            x[k] = (float)v / 255.0f;                      // .float() / 255.0  (:40-41)
        }
        const float* w1 = W + off_w(1);
        const float* b1 = w1 + NF * 4;
#pragma unroll 4
        for (int n = n0; n < n0 + 16; ++n) {
            float s = b1[n];
#pragma unroll
            for (int k = 0; k < 4; ++k) s = __builtin_fmaf(w1[n * 4 + k], x[k], s);
            act[r * PITCH + n] = s > 0.0f ? s : 0.0f;
        }
    }
    __syncthreads();

    // ---- dense layers 2..5: act[:, K:K+64] = relu(act[:, :K] . W^T + b), K = 64, 128, 192, 256
    //      wave w computes the 32x32 block (rows (w&1)*32.., neurons (w>>1)*32..) with v_mfma_f32_32x32x2_f32:
    //      A[i][k] from lane i + 32 k, B[k][j] from lane j + 32 k, D[8(r/4) + 4(lane/32) + r%4][lane%32] in register r
    const int mrow = (wave & 1) * 32, ncol = (wave >> 1) * 32;
#pragma unroll 1
    for (int layer = 2; layer <= 5; ++layer) {
        const int K = (layer - 1) * NF;
        const float* w = W + off_w(layer);
        const float* b = w + NF * K;
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const float* ap = act + (mrow + (lane & 31)) * PITCH + (lane >> 5);
        const float* bp = w + (size_t)(ncol + (lane & 31)) * K + (lane >> 5);
#pragma unroll 1
        for (int k0 = 0; k0 < K; k0 += 16) {
            float av[8], bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                av[s] = ap[k0 + 2 * s];
                bv[s] = bp[k0 + 2 * s];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        }
        const int n = ncol + (lane & 31);
        const float bias = b[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            const float s = acc[r] + bias;
            act[(mrow + i) * PITCH + K + n] = s > 0.0f ? s : 0.0f;
        }
        __syncthreads();
    }

    // ---- conv6 (320 -> outC) + tanh + quantisation: thread = (row, channel)
    if (tid < ROWS * outC) {
        const int r = tid / outC, c = tid - r * outC;
        const float* w6 = W + off_w(6) + c * ACT;
        float s = (W + off_w(6) + outC * ACT)[c];
        const float* a = act + r * PITCH;
#pragma unroll 8
        for (int k = 0; k < ACT; ++k) s = __builtin_fmaf(w6[k], a[k], s);
        const float y = tanhf(s);
        const int e = row0 + r;
        if (e < n_entries) {
            float q = fminf(fmaxf(y, -1.0f), 1.0f) * 127.0f;            // round(clamp(y, -1, 1) * 127)  (:117-119)
            lut[(size_t)e * outC + c] = (int8_t)__builtin_rintf(q);     // torch.round: half to even
            if (yout) yout[(size_t)e * outC + c] = y;
        }
    }
}
}  // namespace transfer

size_t srnet_weight_floats(int outC) { return (size_t)transfer::off_w(6) + (size_t)outC * transfer::ACT + outC; }

int launch_srnet_to_lut(const float* weights, int outC, int interval, int8_t* lut, float* y, hipStream_t st) {
    if (outC < 1 || outC > 4 || interval < 1 || interval > 7) return LERF_EUNSUPPORTED;
    const int L = (1 << (8 - interval)) + 1;
    const long long n = (long long)L * L * L * L;
    if (n > 0x7FFFFFFF) return LERF_EUNSUPPORTED;
    const int blocks = (int)((n + transfer::ROWS - 1) / transfer::ROWS);
    const int lds = transfer::ROWS * transfer::PITCH * (int)sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(transfer::srnet_lut_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
        return LERF_ELAUNCH;
    hipLaunchKernelGGL(transfer::srnet_lut_kernel, dim3(blocks), dim3(256), lds, st, weights, outC, interval, L, (int)n, lut, y);
    return LERF_OK;
}

}  // namespace lerf
