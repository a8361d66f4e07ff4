// Tile-fused uint8 SR path for MI355X (gfx950): host-side dispatch and the LUT pack.  The kernels live in
// lerf_fused_impl.h, instantiated once per channel count: this file holds the RGB instance with the SPECIALISED kernels
// of the published configuration (modes "sct" / "sct", scale factors below 4.9); lerf_fused_g3.hip / _c1.hip / _c4.hip
// hold the general kernels (any 1..4 sampling patterns per stage, scale factors up to 8, frames of different sizes in
// one launch) for 3, 1 and 4 channels.
//
// Reference path being replaced: eltr._worker, resample/eval_lut_sr.py:541-665
// (FourSimplexInterpFaster :24-470 x 24 passes, SteeringGaussianResize2dNumpy /
// AmplifiedLinearResize2dNumpy, resize_right/resize_right2d_numpy.py:142-282).
#define LERF_FUSED_NS fused
#define LERF_FUSED_CH 3
#define LERF_FUSED_WITH_WARP 1      // this instance also carries the tile-fused warp (launch_warp_fused)
#include "lerf_fused_impl.h"

namespace lerf {

static bool modes_ok(const char* m, int n) {
    if (n < 1 || n > fused::MAXM) return false;
    for (int i = 0; i < n; ++i)
        if (m[i] != 's' && m[i] != 'd' && m[i] != 'y' && m[i] != 'c' && m[i] != 't') return false;
    return true;
}
static bool luts_packable(const lerf_luts_t* L) {
    return L && (L->oC == 1 || L->oC == 3) && modes_ok(L->modes1, L->n_modes1) && modes_ok(L->modes2, L->n_modes2);
}

// the specialised kernels of this file: RGB, modes "sct" / "sct", one frame size, scale < 4.9
static bool fused_fast(const FusedArgs& a) {
    const lerf_luts_t* L = a.luts;
    if (a.C != 3 || a.items != nullptr) return false;
    if (a.flags & LERF_GEO_FORCE_GENERAL) return false;                             // diagnostic: A/B of the two kernel families
    if (L->n_modes1 != 3 || L->n_modes2 != 3 || memcmp(L->modes1, "sct", 3) != 0 || memcmp(L->modes2, "sct", 3) != 0) return false;
    return (int64_t)a.oH <= 4 * (int64_t)a.H + 8 && (int64_t)a.oW <= 4 * (int64_t)a.W + 8;      // geometry staging
}

static bool shape_ok(int H, int W, int oH, int oW) {
    if (oH < H || oW < W) return false;                                                   // up-sampling only
    return (int64_t)oH <= 8 * (int64_t)H + 8 && (int64_t)oW <= 8 * (int64_t)W + 8;         // large-geometry staging
}

bool fused_supported(const FusedArgs& a) {
    const lerf_luts_t* L = a.luts;
    if (!luts_packable(L) || !L->fused_pack) return false;
    if ((a.C != 1 && a.C != 3 && a.C != 4) || (a.S != 2 && a.S != 4)) return false;
    if (a.kind == LERF_KIND_LINEAR && a.S != 2) return false;
    if (a.kind == LERF_KIND_GAUSS && !(a.max_sigma <= s3::kNoShiftMaxSigma)) return false;   // float64 direct kernel (lerf_stage3.h)
    if (a.pad_mode < LERF_PAD_CONSTANT || a.pad_mode > LERF_PAD_WRAP) return false;
    const bool roi = a.roi_h > 0 && a.roi_w > 0;
    // wrap padding: the far-side pixels come from the stage-1 output of the two-launch path -- no workspace, or a workspace the
    // caller told the launch not to use (LERF_GEO_SINGLE_LAUNCH), means the direct kernels
    if (a.pad_mode == LERF_PAD_WRAP && (a.workspace == nullptr || (a.flags & LERF_GEO_SINGLE_LAUNCH) || roi)) return false;
    if (a.items != nullptr) {
        if (a.n_items < 1 || roi) return false;
        for (int i = 0; i < a.n_items; ++i)
            if (a.items[i].H < 1 || a.items[i].W < 1 || !shape_ok(a.items[i].H, a.items[i].W, a.items[i].oH, a.items[i].oW)) return false;
        return true;
    }
    if (roi && (a.roi_y < 0 || a.roi_x < 0 || a.roi_y + a.roi_h > a.H || a.roi_x + a.roi_w > a.W)) return false;
    // a region of interest lists the outputs of the REGION (a block, or one part of it: dist.block_parts): the up-sampling test is the region's
    return shape_ok(roi ? a.roi_h : a.H, roi ? a.roi_w : a.W, a.oH, a.oW);
}

size_t fused_workspace_bytes(const FusedArgs& a) {
    if (a.items != nullptr) {
        size_t t = 0;
        for (int i = 0; i < a.n_items; ++i) t += ((size_t)a.items[i].H * a.items[i].W * a.C + 15) / 16 * 16;
        return t;
    }
    return (size_t)a.n * (((size_t)a.H * a.W * a.C + 15) / 16 * 16);
}

// Tile rows of an RGB launch: 64 unless the launch is too small to fill the chip with 64-row tiles -- then 32 or 16 rows (general
// kernels), whichever first gives every CU something to do.  A 256 x 256 frame: 16 tiles of 64 rows on 256 CUs (0.158 ms) ->
// 64 tiles of 16 rows.  Larger launches stay on 64 rows: a tile's fixed costs make the small tiles slower per pixel.
// LERF_GEO_TILE_ROWS_* force a height (tests, A/B).
static int tile_rows_for(const FusedArgs& a) {
    if (a.flags & LERF_GEO_TILE_ROWS_64) return 64;
    if (a.flags & LERF_GEO_TILE_ROWS_32) return 32;
    if (a.flags & LERF_GEO_TILE_ROWS_16) return 16;
    auto tiles = [&](int th) {
        int64_t t = 0;
        if (a.items != nullptr) {
            for (int i = 0; i < a.n_items; ++i) t += (int64_t)((a.items[i].H + th - 1) / th) * ((a.items[i].W + 63) / 64);
            return t;
        }
        const bool roi = a.roi_h > 0 && a.roi_w > 0;
        return (int64_t)a.n * (((roi ? a.roi_h : a.H) + th - 1) / th) * (((roi ? a.roi_w : a.W) + 63) / 64);
    };
    constexpr int64_t kEnough = 192;                 // three quarters of the 256 CUs busy: the larger tile wins from here on
    if (tiles(64) >= kEnough) return 64;
    if (tiles(32) >= kEnough) return 32;
    return tiles(32) >= 96 ? 32 : 16;               // (very small launches: the most workgroups)
}

int launch_sr_fused(const FusedArgs& a, hipStream_t st) {
    if (a.C == 1) return launch_sr_fused_c1(a, st);
    if (a.C == 4) return launch_sr_fused_c4(a, st);
    if (a.C != 3) return LERF_EUNSUPPORTED;
    const int th = tile_rows_for(a);
    if (th == 32) return launch_sr_fused_h32(a, st);
    if (th == 16) return launch_sr_fused_h16(a, st);
    return fused_fast(a) ? fused::launch_sr<false>(a, st) : launch_sr_fused_g3(a, st);
}

// the whole warp path of a batch of RGB frames that share one homography: s1_kernel, then stage 2 + the warp per source tile
// (sr_fused_kernel<.., LERF_FUSED_WARP>): no packed maps travel through HBM
bool warp_fused_supported(const FusedArgs& a) {
    const lerf_luts_t* L = a.luts;
    if (!luts_packable(L) || !L->fused_pack || a.C != 3 || a.S != 2 || a.items != nullptr || !a.wgeo) return false;
    if (L->n_modes1 != 3 || L->n_modes2 != 3 || memcmp(L->modes1, "sct", 3) != 0 || memcmp(L->modes2, "sct", 3) != 0) return false;
    if ((a.kind == LERF_KIND_GAUSS && L->oC != 3) || (a.kind == LERF_KIND_LINEAR && L->oC != 1)) return false;
    if (a.kind != LERF_KIND_GAUSS && a.kind != LERF_KIND_LINEAR) return false;
    if (a.kind == LERF_KIND_GAUSS && !(a.max_sigma <= s3::kNoShiftMaxSigma)) return false;
    if (a.wgeo->pad_mode != LERF_PAD_CONSTANT || a.wgeo->S != 2) return false;
    return (int64_t)a.wgeo->oH * a.wgeo->oW < (1ll << 26);                             // the tile boxes' index division
}

int launch_warp_fused(const FusedArgs& a, hipStream_t st) {
    if (!warp_fused_supported(a)) return LERF_EUNSUPPORTED;
    return fused::launch_warp(a, st);
}

// stages 1+2 only: packed (hq0,hq1,hq2,feat) dwords per pixel-channel
bool fused_stages_supported(const FusedArgs& a) {
    const lerf_luts_t* L = a.luts;
    if (!luts_packable(L) || !L->fused_pack) return false;
    return a.C == 1 || a.C == 3 || a.C == 4;
}

int launch_stages_fused(const FusedArgs& a, hipStream_t st) {
    if (a.C == 1) return launch_stages_fused_c1(a, st);
    if (a.C == 4) return launch_stages_fused_c4(a, st);
    if (a.C != 3) return LERF_EUNSUPPORTED;
    const int th = tile_rows_for(a);
    if (th == 32) return launch_stages_fused_h32(a, st);
    if (th == 16) return launch_stages_fused_h16(a, st);
    return fused_fast(a) ? fused::launch_stages<false>(a, st) : launch_stages_fused_g3(a, st);
}

// fused LUT pack: [n1 x LUT_PAD int8 stage-1 LUTs][2 n2 stage-2 LUTs], n1 = len(modes1), n2 = len(modes2); stage 2 as
//   oC == 3: 2 n2 x NBIN pieces of PIECE_BYTES (dwords of biased bytes e0+128 | e2+128 << 16 | e1+128 << 24),
//            order modes2[0] r0, modes2[0] r1, modes2[1] r0, ...
//   oC == 1: 2 n2 x LUT_PAD int8
size_t fused_lutpack_bytes(const lerf_luts_t* L) {
    if (!luts_packable(L)) return 0;
    return (size_t)L->n_modes1 * fused::LUT_PAD +
           (size_t)2 * L->n_modes2 * (L->oC == 3 ? (size_t)fused::NBIN * fused::PIECE_BYTES : (size_t)fused::LUT_PAD);
}

// stage-2 LUT (3 channels) -> NBIN pieces of biased-byte dwords (e0+128 | e2+128 << 16 | e1+128 << 24); piece b = LUT
// entries [bin_lo(b) * kStrideA, + PIECE_ENTRIES) (zero beyond the LUT); every 1-KiB block pre-permuted for the
// 16-byte-load / ds_write_addtid_b32 transfer: block dword 4L+c holds logical dword 64c+L.
__global__ void pack_s2_pieces_kernel(const int8_t* __restrict__ src, uint32_t* __restrict__ dst) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;      // dword inside a piece
    const int b = blockIdx.y;
    if (p >= fused::PIECE_BYTES / 4) return;
    const int B = p >> 8, r = p & 255, L = r >> 2, c = r & 3;
    const int J = B * 256 + c * 64 + L;
    const int i = fused::bin_lo(b) * kStrideA + J;
    uint32_t d = 0;
    if (J < fused::PIECE_ENTRIES && i < LERF_LUT_ENTRIES) {
        d = (uint32_t)((int)src[i * 3 + 0] + 128) | ((uint32_t)((int)src[i * 3 + 2] + 128) << 16) |
            ((uint32_t)((int)src[i * 3 + 1] + 128) << 24);
    }
    dst[(size_t)b * (fused::PIECE_BYTES / 4) + p] = d;
}

__global__ void pack_bytes_kernel(const int8_t* __restrict__ src, int8_t* __restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < fused::LUT_PAD) dst[i] = i < LERF_LUT_ENTRIES ? src[i] : (int8_t)0;
}
int fused_lutpack_build(const lerf_luts_t* L, void* buf, hipStream_t st) {
    if (!luts_packable(L)) return LERF_EUNSUPPORTED;
    uint8_t* base = (uint8_t*)buf;
    dim3 block(256), grid((fused::LUT_PAD + 255) / 256);
    for (int m = 0; m < L->n_modes1; ++m) {
        if (!L->s1[m]) return LERF_EINVAL;
        hipLaunchKernelGGL(pack_bytes_kernel, grid, block, 0, st, L->s1[m], (int8_t*)(base + (size_t)m * fused::LUT_PAD));
    }
    uint8_t* s2 = base + (size_t)L->n_modes1 * fused::LUT_PAD;
    for (int m = 0; m < L->n_modes2; ++m)
        for (int r = 0; r < 2; ++r) {
            if (!L->s2[m][r]) return LERF_EINVAL;
            int l = m * 2 + r;
            if (L->oC == 3)
                hipLaunchKernelGGL(pack_s2_pieces_kernel, dim3((fused::PIECE_BYTES / 4 + 255) / 256, fused::NBIN), block, 0, st,
                                   L->s2[m][r], (uint32_t*)(s2 + (size_t)l * fused::NBIN * fused::PIECE_BYTES));
            else
                hipLaunchKernelGGL(pack_bytes_kernel, grid, block, 0, st, L->s2[m][r],
                                   (int8_t*)(s2 + (size_t)l * fused::LUT_PAD));
        }
    return LERF_OK;
}

}  // namespace lerf
