// Internal launch interfaces between lerf_api.hip and the kernel files.
#pragma once

#include "lerf_common.h"

namespace lerf {

struct Offsets4 {
    int8_t dy[4];
    int8_t dx[4];
};

// one LUT stage: per mode two LUTs (rotation parity 0 / 1) and 4 rotated patterns
struct StageLuts {
    int n_modes;
    const int8_t* lut[LERF_MAX_MODES][2];
    Offsets4 off[LERF_MAX_MODES][4];
};

struct ResizeArgs {
    const void* feat; int in_dtype; int64_t fy, fx, fc;
    const void* h[3]; int h_dtype; int64_t hy, hx, hc;
    int H, W, C, S, oH, oW;
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;
    int kind; double max_sigma;
    void* out; int out_dtype; int64_t oy, ox, oc;
    int pad_mode;                    // LERF_PAD_* of the image operand
};

struct WarpGeo {
    int S, oH, oW;
    double minv[9];
    int pad_r_lo, pad_r_hi, pad_c_lo, pad_c_hi;
    int pad_mode;                    // LERF_PAD_* of the image operand
    int oy0, ox0;                    // the launch covers the output rectangle that starts at (oy0, ox0) (lerf_warp_geo_t.out_y0 / out_x0)
};

struct WarpArgs {
    const void* feat; int in_dtype; int64_t fy, fx, fc;
    const void* h[3]; int h_dtype; int64_t hy, hx, hc;
    int H, W, C;
    WarpGeo geo;
    int kind; double max_sigma;
    void* out; int out_dtype; int64_t oy, ox, oc;
};

int launch_lut_interp(const void* img, int in_dtype, int64_t sy, int64_t sx, int64_t sc, int img_h, int img_w, int C,
                      int h, int w, Offsets4 off, const int8_t* lut, int oC, int interval, void* out, int out_dtype, int64_t oy,
                      int64_t ox, int64_t ocs, int flags, hipStream_t st);
int launch_numer_epilogue(const int16_t* acc, int64_t n, int interval, const lerf_epi_op_t* ops, int n_ops, float* out, hipStream_t st);
int launch_lut_stage(const uint8_t* img, int64_t sy, int64_t sx, int64_t sc, int H, int W, int C,
                     const StageLuts& luts, int oC, int div, int bias,
                     uint8_t* out, int64_t oy, int64_t ox, int64_t ocs, hipStream_t st);
int launch_resize(const ResizeArgs& a, hipStream_t st);
int launch_warp(const WarpArgs& a, hipStream_t st);

// lerf_fused*.hip (lerf_fused_impl.h)
struct FusedItem {                        // one frame of a ragged launch: its own size and geometry tables
    const uint8_t* img; uint8_t* out; void* emit;
    int H, W, oH, oW;
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;
};
struct FusedArgs {
    const uint8_t* img; int64_t in_sn; int n, H, W, C;
    const lerf_luts_t* luts;
    int S, oH, oW;
    const int* left_r; const float* dis_r; const int* left_c; const float* dis_c;
    const double* dis_r64; const double* dis_c64;
    int kind; float max_sigma;
    uint8_t* out; int64_t out_sn;
    void* workspace; size_t workspace_bytes;
    void* emit; int64_t emit_sn;          // launch_stages_fused: packed dwords out
    int roi_y, roi_x, roi_h, roi_w;       // lerf_sr_geo_t region of interest (roi_h = 0: whole frame)
    int tq_cap;                           // lerf_sr_geo_t.tie_queue_cap
    int pad_mode;                         // LERF_PAD_* of the image operand
    bool host_input;                      // img is (pinned) host memory: the kernels must not re-read pixels from it
    int flags;                            // lerf_sr_geo_t.flags (LERF_GEO_*)
    int out_pitch;                        // lerf_sr_geo_t.out_row_pitch (bytes; 0 = dense)
    const FusedItem* items; int n_items;  // ragged launch (general kernels): frames of different sizes; img/out/H/W/... above unused
    const WarpGeo* wgeo; const int32_t* wboxes;   // launch_warp_fused: the homography and the per-tile output boxes (device)
};
bool fused_supported(const FusedArgs& a);          // some tile-fused kernel covers the configuration
size_t fused_workspace_bytes(const FusedArgs& a);  // what the two-launch path parks in the workspace
size_t fused_lutpack_bytes(const lerf_luts_t* L);
int fused_lutpack_build(const lerf_luts_t* L, void* buf, hipStream_t st);
int launch_sr_fused(const FusedArgs& a, hipStream_t st);
bool fused_stages_supported(const FusedArgs& a);
int launch_stages_fused(const FusedArgs& a, hipStream_t st);
bool warp_fused_supported(const FusedArgs& a);
int launch_warp_fused(const FusedArgs& a, hipStream_t st);    // stage 1 -> workspace, then stage 2 + the warp per source tile
// the general kernels (any patterns, scale <= 8, ragged frames) per channel count, one translation unit each
int launch_sr_fused_g3(const FusedArgs& a, hipStream_t st);
int launch_stages_fused_g3(const FusedArgs& a, hipStream_t st);
int launch_sr_fused_h32(const FusedArgs& a, hipStream_t st);      // RGB, 32- and 16-row tiles (small launches)
int launch_stages_fused_h32(const FusedArgs& a, hipStream_t st);
int launch_sr_fused_h16(const FusedArgs& a, hipStream_t st);
int launch_stages_fused_h16(const FusedArgs& a, hipStream_t st);
int launch_sr_fused_c1(const FusedArgs& a, hipStream_t st);
int launch_stages_fused_c1(const FusedArgs& a, hipStream_t st);
int launch_sr_fused_c4(const FusedArgs& a, hipStream_t st);
int launch_stages_fused_c4(const FusedArgs& a, hipStream_t st);
// lerf_transfer.hip
size_t srnet_weight_floats(int outC);
int launch_srnet_to_lut(const float* weights, int outC, int interval, int8_t* lut, float* y, hipStream_t st);

// lerf_ubench.hip
int launch_ubench_lds_gather(int pattern, int iters, int blocks, uint32_t* sink, hipStream_t st);

int launch_unpack_stages(const uint32_t* packed, int64_t n_pxch, int oC, uint8_t* feat, uint8_t* hq, hipStream_t st);
int launch_warp_packed(const uint32_t* packed, int64_t packed_sn, int n, int H, int W, int C, const WarpGeo& geo, int kind,
                       float max_sigma, void* out, int out_dtype, int64_t oy, int64_t ox, int64_t oc, int64_t out_sn, hipStream_t st);
int launch_rect_copy(uint8_t* frames, int n, int fh, int fw, int C, uint8_t* staging, const lerf_rect_t* rects, int n_rects,
                     int to_staging, hipStream_t st);

}  // namespace lerf
