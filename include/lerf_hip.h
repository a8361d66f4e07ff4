/*
 * lerf_hip.h -- C ABI of liblerf_hip.so: the MI355X (gfx950) implementation of
 * the LeRF LUT resampling hot path.
 *
 * The reference (ddlee-cn/LeRF-PyTorch) is pure Python and has no FFI; the
 * boundary a replacement has to offer is its Python class/function API
 * (SURVEY.md section 8b).  This header is what the Python mirror of that API
 * (the lerf-pytorch_amd Python package) binds through ctypes.  Every entry point cites the
 * reference code it replaces (paths relative to the upstream repository).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in the signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = default stream);
 *   - every function returns 0 on success or a negative LERF_E* code -- no
 *     exceptions cross the ABI, nothing is allocated, there is no mutable
 *     global state (per-device "kernel attribute set" flags apart); the caller
 *     owns every buffer, LUT buffers are borrowed read-only;
 *   - device entry points only enqueue work on `stream`; they never
 *     synchronise with the host;
 *   - re-entrant: concurrency = different streams.
 *   - image operands are described by a base pointer plus ELEMENT strides
 *     (sy, sx, sc) so that HWC-interleaved uint8 frames and planar CHW float
 *     tensors go through the same kernels.
 */
#ifndef LERF_HIP_H
#define LERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LERF_ABI_VERSION 7
#define LERF_MAX_MODES 5          /* s, c, t, d, y  (resample/eval_lut_sr.py:12-18) */
#define LERF_LUT_ENTRIES 83521    /* 17^4, interval = 4 (resample/eval_lut_sr.py:27-28) */
#define LERF_MAX_SUPPORT 8

enum {
    LERF_OK = 0,
    LERF_EINVAL = -1,        /* bad argument (null pointer, size <= 0, unknown mode ...) */
    LERF_EUNSUPPORTED = -2,  /* valid request this build has no kernel for */
    LERF_ELAUNCH = -3,       /* HIP reported a launch error */
    LERF_ENODEVICE = -4      /* no usable gfx950 device */
};

enum { LERF_U8 = 0, LERF_F32 = 1, LERF_F64 = 2, LERF_I16 = 3 };
enum {
    LERF_KIND_GAUSS = 0,     /* steering Gaussian, 3 hyper maps  (LeRF-G) */
    LERF_KIND_LINEAR = 1,    /* amplified linear, 1 hyper map    (LeRF-L) */
    LERF_KIND_NEAREST = 2,   /* box2d                              (resize_right/interp_methods.py:67-70, 83-85) */
    LERF_KIND_CUBIC = 3,     /* cubic2d   (:35-43, 73-75)   -- fixed kernels: lerf_warp only, no hyper maps */
    LERF_KIND_BILINEAR = 4,  /* linear2d  (:60-64, 78-80) */
    LERF_KIND_LANCZOS2 = 5,  /* lanczos2d (:46-50, 88-90) */
    LERF_KIND_LANCZOS3 = 6   /* lanczos3d (:53-57, 93-95) */
};

/* Padding rule of the IMAGE operand of stage 3 (the hyper maps are always edge-padded, :172-174): the `pad_mode` argument of
 * the reference's resampler classes, np.pad / F.pad names in brackets (resize_right2d_numpy.py:143,208; _torch.py:189). */
enum {
    LERF_PAD_CONSTANT = 0,   /* zeros ["constant"] -- the default of every class and the only mode of the uint8 paths */
    LERF_PAD_EDGE = 1,       /* ["edge" / "replicate"] */
    LERF_PAD_REFLECT = 2,    /* ["reflect"] */
    LERF_PAD_SYMMETRIC = 3,  /* ["symmetric"] (numpy only) */
    LERF_PAD_WRAP = 4        /* ["wrap" / "circular"] */
};

typedef struct {
    const void* ptr;      /* device pointer */
    int dtype;            /* LERF_U8 / LERF_F32 / LERF_F64 */
    int64_t sy, sx, sc;   /* element strides of row, column, channel */
} lerf_plane_t;

typedef struct {
    void* ptr;
    int dtype;
    int64_t sy, sx, sc;
} lerf_mplane_t;

/* The LUT set of one model, as the reference loads it
 * (resample/eval_lut_sr.py:750-775): int8, C order [17^4][oC]. */
typedef struct {
    int n_modes1;                               /* len(opt.modes)  */
    int n_modes2;                               /* len(opt.modes2) */
    char modes1[LERF_MAX_MODES];                /* e.g. 's','c','t' */
    char modes2[LERF_MAX_MODES];
    int oC;                                     /* 3 = LeRF-G (rho, sigma_x, sigma_y), 1 = LeRF-L (alpha) */
    const int8_t* s1[LERF_MAX_MODES];           /* device, [17^4]      LUT_s1_<mode>r0 */
    const int8_t* s2[LERF_MAX_MODES][2];        /* device, [17^4][oC]  LUT_s2_<mode>r{0,1} */
    const void* fused_pack;                     /* optional (may be NULL): lerf_fused_lutpack_build output */
} lerf_luts_t;

/* Separable SR geometry: the 1-D content of Resize2dNumpy.get_distance's dense
 * maps (resize_right/resize_right2d_numpy.py:106-140); build with
 * lerf_sr_axis_tables and upload. */
typedef struct {
    int S;                   /* support size (2 in every published result, 4 = class default) */
    int out_h, out_w;
    const int32_t* left_r;   /* device [out_h]    first source row of the support (unpadded coords) */
    const float* dis_r;      /* device [out_h*S]  row distances: uint8 outputs (float32 production arithmetic) */
    const int32_t* left_c;   /* device [out_w] */
    const float* dis_c;      /* device [out_w*S] */
    const double* dis_r64;   /* device [out_h*S]: REQUIRED for LERF_F32 and LERF_F64 outputs of lerf_resize (both are evaluated */
    const double* dis_c64;   /* in float64, LERF_EINVAL without them); optional for uint8 outputs, where they arm the tie guard */
    int pad_mode;            /* LERF_PAD_* of the image operand; non-constant modes: float outputs of lerf_resize only */
    /* ---- ABI 4 (zero = the behaviour of ABI 3) */
    int tie_queue_cap;       /* test hook of the tile-fused kernel: outputs within 1.5e-4 of a rounding tie are queued per tile and
                              * re-evaluated in float64 behind the tile's task loop; 0 = the default capacity (2048 per tile), n > 0 =
                              * n entries, < 0 = no queue (every tie is evaluated inside the loop).  Lets the parity tests drive the
                              * overflow path; carried per call, the library keeps no state */
    int roi_y, roi_x;        /* LR region the tiles of lerf_sr_fused_u8 are laid over (origin and extent in LR pixels); roi_h = 0 or */
    int roi_h, roi_w;        /* roi_w = 0: the whole frame.  A rank of a 2-D block partition passes its OWNED block here and the block
                              * plus halo as the frame: halo pixels then only ever serve as tile halos (255 instead of 288 tiles for a
                              * 1080x960 block of a 2160x3840 frame).  OWNERSHIP RULE (ABI 5, exact): the output tables must list
                              * exactly the outputs whose support CENTRE lies in the region, left + S/2 in [roi_y, roi_y + roi_h)
                              * for rows (columns alike) -- i.e. left in [roi_y - S/2, roi_y + roi_h - S/2); outputs beyond a TRUE
                              * frame border (left + S/2 < 0 or >= H) belong to the region that touches that border.  This is the
                              * slicing rule of lerf-pytorch_amd/dist.py (BlockPlan / StripPlan, `check_support`).  A tile looks the
                              * hyper-parameters up on [tile - S/2, tile + T + S/2 - 1): an output whose support merely STARTS in
                              * the last row / column of the region (left = roi_y + roi_h - 1; accepted by ABI 4's wording) reads a
                              * position no tile of this launch fills.  The library cannot check device tables.  With a workspace
                              * the region takes the two-launch path too: stage 1 runs once per pixel over the region widened by
                              * 3 + S/2 pixels (ABI 5; ABI 4 recomputed stage 1 on every tile's halo for regions) */
    /* ---- ABI 5 (zero = the behaviour of ABI 4) */
    int flags;               /* LERF_GEO_* bits, per call (the library reads no environment variables and keeps no state) */
    int out_row_pitch;       /* lerf_sr_fused_u8: BYTES between output rows; 0 = dense rows of out_w * C bytes.  >= out_w * C.  A rank of
                              * a block partition owns 1919 or 1921 output columns at x2: with rows padded to a multiple of 16 bytes its
                              * tiles keep the aligned store paths (dense 5757-byte rows cost +13 % per launch) */
} lerf_sr_geo_t;
#define LERF_GEO_FORCE_GENERAL 1   /* diagnostic: take the general tile-fused kernels where the specialised ones would serve (A/B runs) */
#define LERF_GEO_SINGLE_LAUNCH 2   /* diagnostic: the single-launch kernel although a workspace is passed (the stamped build keeps its stamps there) */
#define LERF_GEO_X2_TABLES 16      /* the caller vouches that the tables are lerf_sr_axis_tables(scale = 2) on both axes (or row / column slices of
                                    * them).  Reserved: read only by the round-4 persistent-kernel experiment (experiments/r04_persist,
                                    * not in this library); ignored */
#define LERF_GEO_NO_PERSIST 32     /* reserved (same experiment); ignored */
#define LERF_GEO_TILE_ROWS_64 64    /* force the tile height of the RGB tile-fused kernels (default: 64 rows, 32 / 16 for launches too small to */
#define LERF_GEO_TILE_ROWS_32 128   /* fill the chip with 64-row tiles -- a 256 x 256 frame runs as 64 tiles of 16 rows); tests and A/B runs */
#define LERF_GEO_TILE_ROWS_16 256
#define LERF_GEO_INPUT_DEVICE 4    /* lerf_sr_fused_u8: the input frames are device memory / pinned host memory (read over PCIe from inside the */
#define LERF_GEO_INPUT_HOST 8      /* kernel, once per pixel); neither bit: the library asks the runtime (one hipPointerGetAttributes per call) */

/* Homography geometry (resize_right/resize_right2d_numpy.py:306-407): evaluated
 * per output pixel on the device in float64; pad_* come from lerf_warp_pads. */
typedef struct {
    int S;
    int out_h, out_w;
    double minv[9];          /* inverse of the 3x3 matrix, row major */
    int pad_r_lo, pad_r_hi, pad_c_lo, pad_c_hi;
    int pad_mode;            /* LERF_PAD_*; non-constant modes: float outputs of lerf_warp only */
    /* ---- ABI 7 (zero = the behaviour of ABI 6): one RECTANGLE of the output from a BAND of the source, for partitions of a warp over
     * ranks (lerf-pytorch_amd/dist.py WarpRowPlan).  out_h / out_w are then the rectangle's size, `out` its first pixel; the pixel
     * (i, j) of the rectangle is the output pixel (out_y0 + i, out_x0 + j) -- the offsets enter the projection as integers, so the
     * float64 arithmetic is the whole frame's, bit for bit -- and pad_* stay those of the WHOLE output (lerf_warp_pads).  The source
     * operands (feat / hyper planes, packed maps) hold the frame's rows from src_y0 on: row r of the frame is row r - src_y0 of the
     * operand; H, W stay the frame's.  The caller guarantees that every tap of the rectangle lies in the rows it passes.
     * lerf_warp_fused_u8 takes whole outputs only (offsets must be zero). */
    int out_y0, out_x0, src_y0;
} lerf_warp_geo_t;

/* ---------------------------------------------------------------- host side */
int lerf_abi_version(void);
const char* lerf_strerror(int code);

/* number of visible HIP devices (does not initialise a context); <0 on error */
int lerf_device_count(void);

/* Rotated sampling offsets of mode in {'s','c','t','d','y'} for rotation r,
 * expressed in the unrotated frame (resample/eval_lut_sr.py:30-81, 549-553, 468).
 * LERF_EINVAL for an unknown mode == the reference's ValueError (:84). */
int lerf_mode_offsets(char mode, int rot, int8_t dy[4], int8_t dx[4]);

/* 1-D SR tables, float64 arithmetic in the reference's exact operation order
 * (resize_right/resize_right2d_numpy.py:70-79, 85-90, 100-104, 131-134).
 * left[n_out], dis64[n_out*S], dis32[n_out*S] (class-preserving float32
 * rounding: the <0 / [0,1] / >1 classes of the linear kernel are kept),
 * pads[2] = {pad_lo, pad_hi}.  dis32 / pads may be NULL. */
int lerf_sr_axis_tables(int n_in, int n_out, double scale, int S,
                        int32_t* left, double* dis64, float* dis32, int32_t* pads);

/* The same tables with the float32 arithmetic of the reference's torch classes
 * (Resize2dTorch.get_distance, resize_right/resize_right2d_torch.py:48-103): bit-equal to the class's
 * field_of_view / dis tensors, including the scales where float32 and float64 put a support boundary on different
 * sides (x3, S=2).  pads may be NULL. */
int lerf_sr_axis_tables_f32(int n_in, int n_out, double scale, int S, int32_t* left, float* dis32, int32_t* pads);

/* ceil(scale * n_in) (resize_right/resize_right2d_numpy.py:41-45) */
int lerf_out_size(int n_in, double scale);

/* 3x3 inverse (adjugate).  Callers that need bit parity with the reference pass
 * np.linalg.inv's result instead (resize_right/resize_right2d_numpy.py:327). */
int lerf_invert3x3(const double m[9], double out[9]);

/* Pad sizes {r_lo, r_hi, c_lo, c_hi} of Warp2dNumpy.calc_pad_sz from the two
 * corner pixels (:363-369), for the INVERSE homography `minv`. */
int lerf_warp_pads(const double minv[9], int in_h, int in_w, int out_h, int out_w, int S,
                   int32_t pads[4]);

/* -------------------------------------------------------------- device side */

/* FourSimplexInterpFaster (resample/eval_lut_sr.py:24-470) without the final
 * rot90 and /q: integer numerators (value*16) of the 4-simplex interpolation
 * of `lut` ([17^4][oC] int8) over 4 pixels sampled at offsets (dy[k], dx[k])
 * from each of the h x w positions of the uint8 image `img` (coordinates are
 * clamped to [0,img_h-1] x [0,img_w-1]).  out: int16 [C][oC][h][w].
 * interval: the LUT's sampling interval (:27-28; q = 2^interval, L = 2^(8-interval) + 1 levels per axis, `lut` has
 * L^4 rows, numerators are value * q); 4 for every shipped LUT, 1..7 accepted. */
int lerf_lut_interp_i16(const lerf_plane_t* img, int img_h, int img_w, int C,
                        int h, int w, const int8_t dy[4], const int8_t dx[4],
                        const int8_t* lut, int oC, int interval, int16_t* out, void* stream);

/* ABI 6.  The same pass with the reference's epilogue inside the store: the image may be LERF_U8 or LERF_F32 (float32 arrays of
 * integer values are what the call sites hand over, resample/eval_lut_sr.py:549-553; other values are rounded half-to-even and
 * clipped to 0..255), `out` is LERF_I16 (numerators, value * q), LERF_F32 or LERF_F64 (VALUES, numerator / q: :469) with SIGNED
 * element strides sy / sx over the h x w positions and sc between the C * oC result planes -- a caller that points `ptr` at the
 * right corner and hands in the strides of a rotated view gets np.rot90(result, rot, [1, 2]) (:464-468) written in place. */
int lerf_lut_interp(const lerf_plane_t* img, int img_h, int img_w, int C, int h, int w, const int8_t dy[4], const int8_t dx[4],
                    const int8_t* lut, int oC, int interval, const lerf_mplane_t* out, void* stream);

/* ABI 7.  lerf_lut_interp with per-call flags (0 = lerf_lut_interp).  For the shipped interval (4) the pass runs in persistent
 * workgroups that keep one byte plane of the LUT in LDS (csrc/lerf_lut_interp.hip) when the launch is large enough to pay for it
 * (>= 65 536 positions), the pattern reaches no further than 3 pixels and C <= 4; the direct kernel (LUT gathered from L1 / L2)
 * serves everything else -- same values either way.
 *   LERF_INTERP_ACCUMULATE  out += result instead of out = result: the call sites' `pred += FourSimplexInterpFaster(...)`
 *                           (resample/eval_lut_sr.py:555, :564, :589, :601) without a second pass over the planes.  The planes
 *                           must hold valid numbers (the first call of a sum runs without the flag).
 *   LERF_INTERP_LDS / LERF_INTERP_DIRECT   tests and A/B runs: insist on one of the two kernels (LERF_EUNSUPPORTED when the LDS
 *                           kernel does not cover the call). */
#define LERF_INTERP_ACCUMULATE 1
#define LERF_INTERP_LDS 2
#define LERF_INTERP_DIRECT 4
#define LERF_INTERP_TILE64 8       /* A/B runs: force the larger (128 x 32) / smaller (128 x 16) tile of the LDS kernel (default: by the launch's tile count) */
#define LERF_INTERP_TILE32 16
#define LERF_INTERP_LUT_PLANAR 32  /* `lut` is oC planes of 83 584 bytes (17^4 entries + padding to 64), plane k = channel k of every entry: the
                                    * LDS kernel's own format (it copies one plane per workgroup; from the reference's [17^4][oC] layout it has to
                                    * read all oC).  Interval 4 only; LERF_EUNSUPPORTED when the LDS kernel does not cover the call (the caller
                                    * repeats it with the interleaved table) */
#define LERF_LUT_PLANE_BYTES 83584
int lerf_lut_interp_ex(const lerf_plane_t* img, int img_h, int img_w, int C, int h, int w, const int8_t dy[4], const int8_t dx[4],
                       const int8_t* lut, int oC, int interval, const lerf_mplane_t* out, int flags, void* stream);

/* ABI 7.  The call sites' epilogue of a LUT stage (resample/eval_lut_sr.py:573-577, 621-628, resample/eval_lut_warp.py:136-140,
 * 185-191), applied to the int16 NUMERATORS a chain of LERF_INTERP_ACCUMULATE passes has summed (value = numerator / 2^interval):
 *     np.round(np.clip(pred / avg_factor + bias, 0, norm)).astype(np.float32)
 * as a program of float64 steps in the caller's order -- each step is numpy's own float64 operation (IEEE division, addition,
 * minimum / maximum, round-half-even), the result is rounded once to float32: bit for bit what numpy returns for the float64
 * array acc / 2^interval.  acc: int16 [n]; out: float32 [n] (device, dense). */
enum { LERF_EPI_DIV = 0, LERF_EPI_MUL = 1, LERF_EPI_ADD = 2, LERF_EPI_CLIP = 3, LERF_EPI_ROUND = 4 };
#define LERF_EPI_MAX_OPS 8
typedef struct { int op; double a, b; } lerf_epi_op_t;   /* DIV / MUL / ADD: operand a; CLIP: [a, b]; ROUND: none */
int lerf_numer_epilogue_f32(const int16_t* acc, int64_t n, int interval, const lerf_epi_op_t* ops, int n_ops, float* out, void* stream);

/* LUT pack for the tile-fused kernel (1..4 modes per stage, any of "sdyct"; oC = 1 or 3): the stage-1
 * LUTs padded to 16-byte multiples, and the stage-2 LUTs as one uint32 per
 * entry holding the oC biased bytes, cut into the pieces the kernel stages (layout in DESIGN.md).  The caller owns
 * `buf` (lerf_fused_lutpack_bytes(luts) bytes of device memory; 0 = this LUT set has no fused kernel) and stores it in
 * lerf_luts_t.fused_pack. */
size_t lerf_fused_lutpack_bytes(const lerf_luts_t* luts);
int lerf_fused_lutpack_build(const lerf_luts_t* luts, void* buf, void* stream);

/* Stages 1+2 of eltr._worker (resample/eval_lut_sr.py:541-628,
 * resample/eval_lut_warp.py:104-191): uint8 image -> feat (uint8, same shape)
 * and hyper numerators hq (uint8, [H][W][C][oC] through `hyper` strides with
 * sc = stride of c and the oC values contiguous).  feat/hyper may alias
 * nothing.  Either output pointer may be NULL to skip writing it. */
int lerf_lut_stages_u8(const lerf_plane_t* img, int H, int W, int C,
                       const lerf_luts_t* luts,
                       const lerf_mplane_t* feat, const lerf_mplane_t* hyper,
                       void* stream);

/* Stage 3, SR: SteeringGaussianResize2dNumpy.resize (:162-223) /
 * AmplifiedLinearResize2dNumpy.resize (:243-282) and their Torch twins
 * (resize_right/resize_right2d_torch.py:154-197, 214-247).
 * feat: uint8 or float32.  hyper[k]: uint8 numerators (h = u8/255) or float32
 * maps in [0,1]; k = rho, sigma_x, sigma_y (gauss) or alpha (linear).
 * out: uint8 (clip(rne)), float32 or float64 (float64 arithmetic).
 * kind NEAREST / CUBIC / BILINEAR / LANCZOS2 / LANCZOS3: the fixed-kernel resize of Resize2dTorch.resize +
 * BicubicResize2dTorch (resize_right2d_torch.py:105-138, interp_methods.py:35-95) on the same geometry;
 * `hyper` is ignored (may be NULL). */
int lerf_resize(const lerf_plane_t* feat, const lerf_plane_t hyper[3],
                int H, int W, int C, const lerf_sr_geo_t* geo,
                int kind, double max_sigma, const lerf_mplane_t* out, void* stream);

/* Stage 3, homography: SteeringGaussianWarp2dNumpy.warp (:516-577),
 * AmplifiedLinearWarp2dNumpy.warp (:597-636), NearestWarp2dNumpy (:460-467,
 * 409-449), the fixed-kernel baselines Bicubic/Bilinear/Lanczos2/Lanczos3Warp2dNumpy
 * (:451-494) and the Torch twins (resize_right2d_torch.py:346-487).
 * Pixels whose weights all vanish are NaN in float outputs (the reference's
 * 0/0) and 0 in uint8 outputs.  If `mask_out` (uint8 [out_h][out_w]) is
 * non-NULL and kind == NEAREST, it receives out == 255 per pixel. */
int lerf_warp(const lerf_plane_t* feat, const lerf_plane_t hyper[3],
              int H, int W, int C, const lerf_warp_geo_t* geo,
              int kind, double max_sigma, const lerf_mplane_t* out, void* stream);

/* Stages 1+2 by the tile-fused kernel (luts->fused_pack set), `n` frames:
 * packed[(y*W + x)*C + c] = hq0 | hq1<<8 | hq2<<16 | feat<<24  (hq1, hq2 = 0 for LeRF-L).
 * Same values as lerf_lut_stages_u8, ~4x faster; feeds lerf_warp_packed / lerf_unpack_stages.
 * workspace: optional device scratch (NULL = none) of `workspace_bytes` bytes: with it stage 1 runs as its own launch
 * without recomputing tile halos, like lerf_sr_fused_u8.  LERF_EINVAL when it is non-NULL and smaller than
 * lerf_sr_fused_workspace_bytes(H, W, C, n). */
int lerf_stages_packed_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C,
                          const lerf_luts_t* luts, uint32_t* packed, int64_t packed_sn,
                          void* workspace, size_t workspace_bytes, void* stream);

/* packed dwords -> feat uint8 [n_pxch] and hq uint8 [n_pxch][oC] (either may be NULL) */
int lerf_unpack_stages(const uint32_t* packed, int64_t n_pxch, int oC, uint8_t* feat, uint8_t* hq, void* stream);

/* lerf_warp (gauss / linear) reading the packed stage outputs of `n` HWC frames that share one homography (frame
 * strides packed_sn in dwords, out_sn in elements of `out`; n = 1: one frame) in ONE launch; out: uint8 or float32 */
int lerf_warp_packed(const uint32_t* packed, int64_t packed_sn, int n, int H, int W, int C, const lerf_warp_geo_t* geo,
                     int kind, double max_sigma, const lerf_mplane_t* out, int64_t out_sn, void* stream);

/* ABI 7.  The whole warp path of the harness (resample/eval_lut_warp.py:100-222: stage 1, stage 2, SteeringGaussianWarp2dNumpy /
 * AmplifiedLinearWarp2dNumpy.warp, resize_right/resize_right2d_numpy.py:516-636) for `n` RGB frames that share one homography,
 * TILE-FUSED: stage 1 runs once per pixel into the workspace (as in lerf_sr_fused_u8); the second launch runs stage 2 per 64 x 64
 * source tile and evaluates, from the tile's packed stage outputs in LDS, the output pixels that tile OWNS -- those whose 2 x 2
 * support (clamped into the frame) has its last tap row / column inside the tile; pixels projected outside the frame are clipped
 * onto its border like the reference clips its grid (:338-339) and belong to the border tiles.  No packed maps travel through
 * HBM (lerf_stages_packed_u8 + lerf_warp_packed write and re-read 12 bytes per source pixel); same bytes as that path.
 *   lerf_warp_tile_boxes   host: boxes[t] = {i0, i1, j0, j1}, the output rows x columns that bound what tile t (row-major,
 *                          ceil(H / 64) x ceil(W / 64) tiles) owns; one pass over the output per homography.  The caller uploads
 *                          them (int32 [tiles][4]) and passes the device pointer as `tile_boxes`.
 *   workspace              device scratch of at least lerf_sr_fused_workspace_bytes(H, W, C, n) bytes (required).
 * RGB frames, S = 2, the shipped pattern set "sct" / "sct", constant padding, max_sigma <= 13, out_h * out_w < 2^26:
 * lerf_warp_fused_supported says so; everything else: lerf_stages_packed_u8 + lerf_warp_packed. */
int lerf_warp_tile_boxes(const lerf_warp_geo_t* geo, int H, int W, int32_t* boxes);
int lerf_warp_fused_supported(int C, const lerf_luts_t* luts, const lerf_warp_geo_t* geo, int H, int W, int kind, double max_sigma);
int lerf_warp_fused_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C, const lerf_luts_t* luts, const lerf_warp_geo_t* geo,
                       const int32_t* tile_boxes, int kind, double max_sigma, uint8_t* out, int64_t out_sn, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Whole SR path of eltr._worker (resample/eval_lut_sr.py:541-665) for a batch of `n` frames (batch strides
 * in_sn / out_sn in elements): uint8 HWC in -> uint8 HWC out.
 * workspace != NULL (`workspace_bytes` >= lerf_sr_fused_workspace_bytes(H, W, C, n) bytes of device memory, LERF_EINVAL
 * when shorter): TWO launches over the same grid of 64x64 LR tiles; stage 1 runs once per pixel and its
 * uint8 output (C bytes per LR pixel) waits in the workspace, the second launch runs stage 2, the finalisation and stage
 * 3 per tile; the hyper-parameters never leave the CU.
 * workspace == NULL: ONE launch; every tile recomputes stage 1 on its halo (about 4 % slower on a batch, the better
 * choice for a single frame or block that fills the chip once) and nothing but the input, the LUT pack and the output
 * touches HBM.  Configurations outside the tile-fused kernel (see lerf_sr_fused_supported) need the workspace and run
 * the three direct kernels through it. */
size_t lerf_sr_fused_workspace_bytes(int H, int W, int C, int n);
int lerf_sr_fused_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C,
                     const lerf_luts_t* luts, const lerf_sr_geo_t* geo,
                     int kind, double max_sigma,
                     uint8_t* out, int64_t out_sn, void* workspace, size_t workspace_bytes, void* stream);
/* 1 when lerf_sr_fused_u8 would take the tile-fused kernels for this configuration, 0 when it would fall back to the
 * direct kernels (host-side query, no device work) */
int lerf_sr_fused_supported(int C, const lerf_luts_t* luts, const lerf_sr_geo_t* geo, int H, int W, int kind, double max_sigma);

/* Frames of DIFFERENT sizes through ONE launch pair of the general tile-fused kernels (each workgroup finds its frame in a
 * descriptor table that travels in the kernel arguments: 16 frames per launch pair, more frames = more launch pairs): what
 * eltr.run does image by image over a benchmark folder (resample/eval_lut_sr.py:489-512).  Every item carries its own
 * geometry; the support S, pad_mode, tie_queue_cap and flags must be the same for all items (LERF_EINVAL otherwise), an
 * item with a region of interest sends the call item by item through lerf_sr_fused_u8, as does any configuration without
 * a tile-fused kernel.  All frames, tables and the workspace live on ONE device (the caller's current one).  `items` is a
 * HOST array.  workspace: device scratch of at least lerf_sr_ragged_workspace_bytes(items, n, C) bytes (required). */
typedef struct {
    const uint8_t* img;      /* device, dense uint8 [H][W][C] */
    uint8_t* out;            /* device, dense uint8 [geo.out_h][geo.out_w][C] */
    int H, W;
    lerf_sr_geo_t geo;
} lerf_sr_item_t;
size_t lerf_sr_ragged_workspace_bytes(const lerf_sr_item_t* items, int n, int C);
int lerf_sr_fused_ragged_u8(const lerf_sr_item_t* items, int n, int C, const lerf_luts_t* luts, int kind, double max_sigma,
                            void* workspace, size_t workspace_bytes, void* stream);
/* the same for stages 1+2 alone (packed dwords out, see lerf_stages_packed_u8): the warp harness' stage passes over a
 * folder of images (resample/eval_lut_warp.py:100-191) */
typedef struct {
    const uint8_t* img;      /* device, dense uint8 [H][W][C] */
    uint32_t* packed;        /* device, dense uint32 [H][W][C] */
    int H, W;
} lerf_stage_item_t;
size_t lerf_stages_ragged_workspace_bytes(const lerf_stage_item_t* items, int n, int C);
int lerf_stages_packed_ragged_u8(const lerf_stage_item_t* items, int n, int C, const lerf_luts_t* luts,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Halo plumbing of the multi-GPU partitions (lerf-pytorch_amd/dist.py): copies up to LERF_MAX_RECTS rectangles between
 * a batch of dense uint8 frames [n][fh][fw][C] and one contiguous staging buffer, in ONE launch.
 * to_staging = 1: frame rectangles -> staging (pack before the sends); 0: staging -> frame rectangles (unpack after the
 * receives).  Rectangle r covers rows [y, y+h) x columns [x, x+w) of every frame and lives at staging + off (bytes) as a
 * dense [n][h][w][C] block. */
#define LERF_MAX_RECTS 8
typedef struct { int y, x, h, w; int64_t off; } lerf_rect_t;
int lerf_rect_copy_u8(uint8_t* frames, int n, int fh, int fw, int C, uint8_t* staging,
                      const lerf_rect_t* rects, int n_rects, int to_staging, void* stream);

/* ---- evaluation metrics of the reference harness, on the device (uint8 HWC RGB, row pitch in elements).
 * Each call leaves two doubles in `result` (device memory): a sum and a count; no host sync.
 *
 * lerf_metric_y_sse_u8: ingredients of PSNR(_rgb2ycbcr(gt)[:,:,0], _rgb2ycbcr(out)[:,:,0], shave)
 *   (resample/eval_lut_sr.py:741-742, common/utils.py:46-76, 138-151): result[0] = sum over the frame minus a
 *   `shave`-pixel border of (float32(Y_out) - float32(Y_gt))^2, result[1] = number of pixels summed.
 * lerf_metric_ssim_y_u8: cal_ssim(y_gt, y_out) (eval_lut_sr.py:743, common/utils.py:177-206; 11x11 Gaussian
 *   window sigma 1.5, 'valid', float64): result[0] = sum of the SSIM map, result[1] = (H-10)*(W-10).
 * lerf_metric_masked_sse_u8: ingredients of mPSNR(sr, hr, mask) (resample/eval_lut_warp.py:233,
 *   common/utils.py:168-175) over n = H*W*C elements: result[0] = sum (mask*(sr-hr)/255)^2 in float32 steps,
 *   result[1] = sum(mask); mask is 0 / non-zero bytes. */
int lerf_metric_y_sse_u8(const uint8_t* gt, int64_t gt_sy, const uint8_t* out, int64_t out_sy, int H, int W, int shave,
                         double* result, void* stream);
int lerf_metric_ssim_y_u8(const uint8_t* gt, int64_t gt_sy, const uint8_t* out, int64_t out_sy, int H, int W,
                          double* result, void* stream);
int lerf_metric_masked_sse_u8(const uint8_t* sr, const uint8_t* hr, const uint8_t* mask, int64_t n, double* result,
                              void* stream);

/* ---- fine-tuning path: the trainable float32 twin of the LUT pass, SWF2LUT.InterpTorchBatch
 * (resample/model.py:172-385), forward and backward.
 * weight: float32 [17^4][oC] LUT parameters (LUT value = clamp(rne(127 w), -127, 127), :177-179);
 * img: float32 [n_planes][h+bd][w+bd], integer-valued 0..255, already rotated and replicate-padded by the caller
 * (SWF2LUT.predict, :398-431), n_planes = B*C; out / grad_out: float32 [n_planes][oC][h][w]  (= [B][C*oC][h][w]).
 * mode: one of "sdyct"; modes c and t read their LSBs at the 'y' pattern pixels exactly like the reference
 * (:229-232, :240-243).  bd >= the pattern reach (mode_pad_dict: s 1, d 2, y 2, c 3, t 3).
 * Backward = what autograd derives for the reference code: grad_weight (accumulated with float atomics INTO the
 * caller's buffer, so zero it or pass the running .grad) and grad_img (same; through the LSB terms only).  Either
 * gradient pointer may be NULL. */
int lerf_swf2lut_interp_f32(const float* weight, int oC, char mode, const float* img, int n_planes, int h, int w, int bd,
                            float* out, void* stream);
int lerf_swf2lut_interp_bwd_f32(const float* weight, int oC, char mode, const float* img, const float* grad_out,
                                int n_planes, int h, int w, int bd, float* grad_weight, float* grad_img, void* stream);

/* Backward of lerf_resize (kinds GAUSS, LINEAR) on planar float32 maps, as autograd derives it for
 * SteeringGaussianResize2dTorch.resize / AmplifiedLinearResize2dTorch.resize (resize_right2d_torch.py:154-247):
 * feat, h0..h2: float32 [N][H][W] (hyper maps in [0,1]; h1, h2 unused for LINEAR), grad_out: float32
 * [N][out_h][out_w].  grad_feat / grad_h*: float32 [N][H][W], ACCUMULATED into with float atomics (zero them first);
 * any of them may be NULL.  Uses the float32 distance tables of `geo`. */
int lerf_resize_bwd_f32(const float* feat, const float* h0, const float* h1, const float* h2, int N, int H, int W,
                        const lerf_sr_geo_t* geo, int kind, double max_sigma, const float* grad_out, float* grad_feat,
                        float* grad_h0, float* grad_h1, float* grad_h2, void* stream);

/* ---- net -> LUT transfer (resample/transfer_to_lut.py:12-170): one hyper-network of the reference's SRNetsSWF2
 * (resample/model.py:81-99; an SRNet = SRUnit MLP, common/network.py:40-163) evaluated on all L^4 sampled pixel
 * tuples (L = 2^(8-interval) + 1; get_input_tensor :12-42, first pixel = slowest axis) and quantised like :117-119:
 * lut[e][c] = int8(round_half_even(clamp(y, -1, 1) * 127)), C order [L^4][outC] -- the array the reference saves as
 * LUT_<key>.npy (there with two trailing singleton dims, scripts.sh:19-24).
 * weights: device, lerf_srnet_weight_floats(outC) floats: W1[64][4] b1[64] W2[64][64] b2[64] W3[64][128] b3[64]
 * W4[64][192] b4[64] W5[64][256] b5[64] W6[outC][320] b6[outC]  (the state_dict tensors of one SRNet, flattened in
 * module order).  y (optional, device float32 [L^4][outC]): the network outputs before quantisation.
 * The hidden layers run on the matrix cores (float32-input MFMA: exact float32 arithmetic). */
size_t lerf_srnet_weight_floats(int outC);
int lerf_srnet_to_lut(const float* weights, int outC, int interval, int8_t* lut, float* y, void* stream);

/* ---- calibration (bench.py roofline_lds): `workgroups` x 1024 threads, each wave issuing 10 x `iters` ds_read_b32 gathers
 * into a 134-KB LDS table -- pattern 0: random addresses (the rate a data-dependent LUT gather gets), pattern 1:
 * conflict-free.  The caller times the launch (one workgroup per CU: wave-gathers per CU = 160 x iters) and owns `sink`
 * (4 bytes of device memory, practically never written).  Not part of the reference's path: a ruler for it. */
int lerf_ubench_lds_gather(int pattern, int iters, int workgroups, uint32_t* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LERF_HIP_H */
