// Shared host/device helpers for the LeRF HIP kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lerf_hip.h"
#include "lerf_host_geometry.h"

namespace lerf {

constexpr int kL = 17;                 // 2^(8-interval)+1, interval = 4
constexpr int kQ = 16;                 // 2^interval
constexpr int kStrideA = kL * kL * kL; // 4913: first sampled pixel = slowest LUT axis
constexpr int kStrideB = kL * kL;      // 289
constexpr int kStrideC = kL;           // 17
constexpr int kStrideD = 1;

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i runs on XCD i % 8; an affinity, not a guarantee -- nothing
// depends on it but speed).  xcd_contiguous() hands every XCD a CONTIGUOUS eighth of a launch's block sequence instead of
// every eighth block, so that the blocks running side by side on one XCD are neighbours in the frame and what they read
// together meets in that XCD's own 4-MiB L2.  A bijection of [0, total) for any total.
__device__ __forceinline__ int xcd_contiguous(int b, int total) {
    const int x = b & 7, j = b >> 3, q = total >> 3, r = total & 7;
    return x * q + (x < r ? x : r) + j;
}

// round-half-to-even of n/d for n >= 0 (np.round), clipped to [0,255].
__host__ __device__ inline int rne_div_clip255(int n, int d) {
    if (n <= 0) return 0;
    int q = n / d;
    int r = n - q * d;
    int up = (2 * r > d) || ((2 * r == d) && (q & 1));
    q += up;
    return q > 255 ? 255 : q;
}

// The same for the two divisors the pipeline uses (48 = 16*3 in stage 1, 192 = 16*12 in stage 2), branch-free:
// floor(m / d) = (m * 0xAAAB) >> SH exactly for m < 2^17 (0xAAAB / 2^23 exceeds 1/192 by 2.6e-8 relative), and
// rne(m / d) = floor((m + d/2 - 1 + (floor(m / d) & 1)) / d): the parity term sends exact halves to the even side.
template <int D>
__device__ __forceinline__ int rne_div_clip255_c(int n) {
    static_assert(D == 48 || D == 192, "divisors of the LeRF stages");
    constexpr int SH = D == 48 ? 21 : 23;
    const unsigned m = (unsigned)(n > 0 ? n : 0);
    const unsigned k = __umul24(m, 0xAAABu) >> SH;
    const unsigned q = __umul24(m + (unsigned)(D / 2 - 1) + (k & 1u), 0xAAABu) >> SH;
    return (int)(q < 255u ? q : 255u);
}

__device__ __forceinline__ int rne_div_clip255_fast(int n, int d) {
    if (d == 48) return rne_div_clip255_c<48>(n);      // d is a compile-time constant at every call site
    if (d == 192) return rne_div_clip255_c<192>(n);
    return rne_div_clip255(n, d);
}

// compare-exchange, descending
__device__ __forceinline__ void ce_desc(unsigned& a, unsigned& b) {
    unsigned hi = a > b ? a : b;
    unsigned lo = a > b ? b : a;
    a = hi;
    b = lo;
}

// 4-simplex interpolation walk: sorted (LSB<<16 | axis stride) keys.
// Produces the 5 LUT indices and weights (sum of weights = 16).
struct SimplexPath {
    int idx[5];
    int w[5];
};

__device__ __forceinline__ SimplexPath simplex_path(int v0, int v1, int v2, int v3) {
    unsigned k0 = ((unsigned)(v0 & 15) << 16) | (unsigned)kStrideA;
    unsigned k1 = ((unsigned)(v1 & 15) << 16) | (unsigned)kStrideB;
    unsigned k2 = ((unsigned)(v2 & 15) << 16) | (unsigned)kStrideC;
    unsigned k3 = ((unsigned)(v3 & 15) << 16) | (unsigned)kStrideD;
    int base = (v0 >> 4) * kStrideA + (v1 >> 4) * kStrideB + (v2 >> 4) * kStrideC + (v3 >> 4);
    ce_desc(k0, k1);
    ce_desc(k2, k3);
    ce_desc(k0, k2);
    ce_desc(k1, k3);
    ce_desc(k1, k2);
    int f0 = k0 >> 16, f1 = k1 >> 16, f2 = k2 >> 16, f3 = k3 >> 16;
    SimplexPath p;
    p.idx[0] = base;
    p.idx[1] = p.idx[0] + (int)(k0 & 0xffffu);
    p.idx[2] = p.idx[1] + (int)(k1 & 0xffffu);
    p.idx[3] = p.idx[2] + (int)(k2 & 0xffffu);
    p.idx[4] = p.idx[3] + (int)(k3 & 0xffffu);
    p.w[0] = kQ - f0;
    p.w[1] = f0 - f1;
    p.w[2] = f1 - f2;
    p.w[3] = f2 - f3;
    p.w[4] = f3;
    return p;
}


// Launch errors are what hipGetLastError() reports after the launches of an entry point -- but the thread's error state also
// carries what the CALLER's HIP traffic left there (PyTorch polls its events: hipErrorNotReady).  Entry points drop stale
// codes before they launch (clear_stale_error) and do not take "not ready" for a launch failure.
inline void clear_stale_error() { (void)hipGetLastError(); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && e != hipErrorNotReady && getenv("LERF_DEBUG")) fprintf(stderr, "lerf: HIP error %d: %s\n", (int)e, hipGetErrorString(e));
    return (e == hipSuccess || e == hipErrorNotReady) ? LERF_OK : LERF_ELAUNCH;
}

// Source index of padded position i (unpadded coordinates, may lie outside [0, n)) under the padding rule of the IMAGE
// operand: np.pad(input, pad_vec, mode=self.pad_mode) (resize_right/resize_right2d_numpy.py:208, :560) /
// F.pad(input, pad_vec, mode=self.pad_mode) (resize_right2d_torch.py:189, :362).  *zero: the value is 0 (constant).
// reflect / symmetric / wrap follow numpy for any pad width (periodic extension).
__host__ __device__ inline int pad_index(int i, int n, int mode, bool* zero) {
    *zero = false;
    if (i >= 0 && i < n) return i;
    if (mode == 1) return i < 0 ? 0 : n - 1;                          // edge / replicate
    if (mode == 2) {                                                  // reflect (no edge repeat)
        const int p = 2 * (n - 1);
        if (p == 0) return 0;
        const int m = ((i % p) + p) % p;
        return m < n ? m : p - m;
    }
    if (mode == 3) {                                                  // symmetric (edge repeated)
        const int p = 2 * n;
        const int m = ((i % p) + p) % p;
        return m < n ? m : p - 1 - m;
    }
    if (mode == 4) return ((i % n) + n) % n;                          // wrap / circular
    *zero = true;                                                     // constant (0)
    return i < 0 ? 0 : n - 1;
}

}  // namespace lerf
