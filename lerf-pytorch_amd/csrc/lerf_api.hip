// C ABI of liblerf_hip.so (include/lerf_hip.h): argument checking, host-side
// geometry, and dispatch to the kernels.  No allocation, no synchronisation,
// no global state.
#include <math.h>
#include <string.h>

#include "lerf_kernels.h"

using namespace lerf;

namespace {

// every device entry point converts its stream before it launches anything: the place to drop stale error codes
inline hipStream_t as_stream(void* s) {
    clear_stale_error();
    return reinterpret_cast<hipStream_t>(s);
}

inline int check_launch() { return launch_status(); }

inline bool plane_ok(const lerf_plane_t* p) { return p && p->ptr; }

int build_stage_luts(const lerf_luts_t* L, int stage, StageLuts* out) {
    int n = stage == 1 ? L->n_modes1 : L->n_modes2;
    if (n < 1 || n > LERF_MAX_MODES) return LERF_EINVAL;
    out->n_modes = n;
    for (int m = 0; m < n; ++m) {
        char mode = stage == 1 ? L->modes1[m] : L->modes2[m];
        for (int r = 0; r < 4; ++r)
            if (!mode_offsets(mode, r, out->off[m][r].dy, out->off[m][r].dx)) return LERF_EINVAL;
        if (stage == 1) {
            if (!L->s1[m]) return LERF_EINVAL;
            out->lut[m][0] = out->lut[m][1] = L->s1[m];     // stage 1 uses ...r0 for all rotations (:545)
        } else {
            if (!L->s2[m][0] || !L->s2[m][1]) return LERF_EINVAL;
            out->lut[m][0] = L->s2[m][0];                    // r in {0,2} -> r0, {1,3} -> r1 (:582-619)
            out->lut[m][1] = L->s2[m][1];
        }
    }
    return LERF_OK;
}

}  // namespace

extern "C" {

int lerf_abi_version(void) { return LERF_ABI_VERSION; }

const char* lerf_strerror(int code) {
    switch (code) {
        case LERF_OK: return "ok";
        case LERF_EINVAL: return "invalid argument";
        case LERF_EUNSUPPORTED: return "unsupported configuration";
        case LERF_ELAUNCH: return "HIP launch failed";
        case LERF_ENODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

int lerf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return LERF_ENODEVICE;
    return n;
}

int lerf_mode_offsets(char mode, int rot, int8_t dy[4], int8_t dx[4]) {
    if (!dy || !dx) return LERF_EINVAL;
    return mode_offsets(mode, rot, dy, dx) ? LERF_OK : LERF_EINVAL;
}

int lerf_out_size(int n_in, double scale) { return host::out_size(n_in, scale); }

int lerf_sr_axis_tables(int n_in, int n_out, double scale, int S, int32_t* left, double* dis64, float* dis32, int32_t* pads) {
    return host::sr_axis_tables(n_in, n_out, scale, S, left, dis64, dis32, pads);
}

int lerf_sr_axis_tables_f32(int n_in, int n_out, double scale, int S, int32_t* left, float* dis32, int32_t* pads) {
    return host::sr_axis_tables_f32(n_in, n_out, scale, S, left, dis32, pads);
}

int lerf_invert3x3(const double m[9], double out[9]) { return host::invert3x3(m, out); }

int lerf_warp_pads(const double minv[9], int in_h, int in_w, int out_h, int out_w, int S, int32_t pads[4]) {
    return host::warp_pads(minv, in_h, in_w, out_h, out_w, S, pads);
}

int lerf_lut_interp_i16(const lerf_plane_t* img, int img_h, int img_w, int C, int h, int w, const int8_t dy[4],
                        const int8_t dx[4], const int8_t* lut, int oC, int interval, int16_t* out, void* stream) {
    if (!out) return LERF_EINVAL;
    lerf_mplane_t o;
    o.ptr = out; o.dtype = LERF_I16; o.sy = w; o.sx = 1; o.sc = (int64_t)h * w;
    return lerf_lut_interp(img, img_h, img_w, C, h, w, dy, dx, lut, oC, interval, &o, stream);
}

int lerf_lut_interp(const lerf_plane_t* img, int img_h, int img_w, int C, int h, int w, const int8_t dy[4], const int8_t dx[4],
                    const int8_t* lut, int oC, int interval, const lerf_mplane_t* out, void* stream) {
    return lerf_lut_interp_ex(img, img_h, img_w, C, h, w, dy, dx, lut, oC, interval, out, 0, stream);
}

int lerf_lut_interp_ex(const lerf_plane_t* img, int img_h, int img_w, int C, int h, int w, const int8_t dy[4], const int8_t dx[4],
                       const int8_t* lut, int oC, int interval, const lerf_mplane_t* out, int flags, void* stream) {
    if (!plane_ok(img) || (img->dtype != LERF_U8 && img->dtype != LERF_F32) || !lut || !dy || !dx) return LERF_EINVAL;
    if (!out || !out->ptr || (out->dtype != LERF_I16 && out->dtype != LERF_F32 && out->dtype != LERF_F64)) return LERF_EINVAL;
    if (img_h < 1 || img_w < 1 || C < 1 || h < 1 || w < 1) return LERF_EINVAL;
    if (flags & ~(LERF_INTERP_ACCUMULATE | LERF_INTERP_LDS | LERF_INTERP_DIRECT | LERF_INTERP_TILE64 | LERF_INTERP_TILE32 | LERF_INTERP_LUT_PLANAR)) return LERF_EINVAL;
    if ((flags & LERF_INTERP_LDS) && (flags & LERF_INTERP_DIRECT)) return LERF_EINVAL;
    Offsets4 off;
    memcpy(off.dy, dy, 4);
    memcpy(off.dx, dx, 4);
    int rc = launch_lut_interp(img->ptr, img->dtype, img->sy, img->sx, img->sc, img_h, img_w, C, h, w, off, lut, oC, interval,
                               out->ptr, out->dtype, out->sy, out->sx, out->sc, flags, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_numer_epilogue_f32(const int16_t* acc, int64_t n, int interval, const lerf_epi_op_t* ops, int n_ops, float* out, void* stream) {
    if (!acc || !out || n < 1 || (n_ops > 0 && !ops)) return LERF_EINVAL;
    int rc = launch_numer_epilogue(acc, n, interval, ops, n_ops, out, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

size_t lerf_fused_lutpack_bytes(const lerf_luts_t* luts) { return luts ? fused_lutpack_bytes(luts) : 0; }

int lerf_fused_lutpack_build(const lerf_luts_t* luts, void* buf, void* stream) {
    if (!luts || !buf) return LERF_EINVAL;
    if (fused_lutpack_bytes(luts) == 0) return LERF_EUNSUPPORTED;
    int rc = fused_lutpack_build(luts, buf, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_lut_stages_u8(const lerf_plane_t* img, int H, int W, int C, const lerf_luts_t* luts,
                       const lerf_mplane_t* feat, const lerf_mplane_t* hyper, void* stream) {
    if (!plane_ok(img) || img->dtype != LERF_U8 || !luts || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    if (!feat || !feat->ptr || feat->dtype != LERF_U8) return LERF_EINVAL;   // stage 2 reads feat
    if (luts->oC != 1 && luts->oC != 3) return LERF_EUNSUPPORTED;
    StageLuts s1, s2;
    int rc = build_stage_luts(luts, 1, &s1);
    if (rc != LERF_OK) return rc;
    hipStream_t st = as_stream(stream);
    // feat = rne(clip(pred/len(modes), 0, 255)): numerators carry a factor 16          (:565-577)
    rc = launch_lut_stage((const uint8_t*)img->ptr, img->sy, img->sx, img->sc, H, W, C, s1, 1, kQ * luts->n_modes1, 0,
                          (uint8_t*)feat->ptr, feat->sy, feat->sx, feat->sc, st);
    if (rc != LERF_OK) return rc;
    if (hyper && hyper->ptr) {
        if (hyper->dtype != LERF_U8) return LERF_EINVAL;
        rc = build_stage_luts(luts, 2, &s2);
        if (rc != LERF_OK) return rc;
        // hq = rne(clip(pred/(4*len(modes2)) + 127, 0, 255))                              (:621-628)
        rc = launch_lut_stage((const uint8_t*)feat->ptr, feat->sy, feat->sx, feat->sc, H, W, C, s2, luts->oC,
                              kQ * 4 * luts->n_modes2, 127, (uint8_t*)hyper->ptr, hyper->sy, hyper->sx, hyper->sc, st);
        if (rc != LERF_OK) return rc;
    }
    return check_launch();
}

int lerf_resize(const lerf_plane_t* feat, const lerf_plane_t hyper[3], int H, int W, int C, const lerf_sr_geo_t* geo,
                int kind, double max_sigma, const lerf_mplane_t* out, void* stream) {
    if (!plane_ok(feat) || !geo || !out || !out->ptr || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    if (kind < LERF_KIND_GAUSS || kind > LERF_KIND_LANCZOS3) return LERF_EUNSUPPORTED;
    int nh = kind == LERF_KIND_GAUSS ? 3 : (kind == LERF_KIND_LINEAR ? 1 : 0);   // fixed kernels take no hyper maps
    if (nh > 0 && !hyper) return LERF_EINVAL;
    for (int k = 0; k < nh; ++k)
        if (!hyper[k].ptr || hyper[k].dtype != hyper[0].dtype || hyper[k].sy != hyper[0].sy ||
            hyper[k].sx != hyper[0].sx || hyper[k].sc != hyper[0].sc)
            return LERF_EINVAL;
    if (!geo->left_r || !geo->left_c || geo->out_h < 1 || geo->out_w < 1) return LERF_EINVAL;
    ResizeArgs a{};
    a.feat = feat->ptr; a.in_dtype = feat->dtype; a.fy = feat->sy; a.fx = feat->sx; a.fc = feat->sc;
    for (int k = 0; k < 3; ++k) a.h[k] = nh == 0 ? nullptr : (k < nh ? hyper[k].ptr : hyper[0].ptr);
    if (nh > 0) { a.h_dtype = hyper[0].dtype; a.hy = hyper[0].sy; a.hx = hyper[0].sx; a.hc = hyper[0].sc; }
    a.H = H; a.W = W; a.C = C; a.S = geo->S; a.oH = geo->out_h; a.oW = geo->out_w;
    a.left_r = geo->left_r; a.dis_r = geo->dis_r; a.left_c = geo->left_c; a.dis_c = geo->dis_c;
    a.dis_r64 = geo->dis_r64; a.dis_c64 = geo->dis_c64;
    a.kind = kind; a.max_sigma = max_sigma;
    a.out = out->ptr; a.out_dtype = out->dtype; a.oy = out->sy; a.ox = out->sx; a.oc = out->sc;
    a.pad_mode = geo->pad_mode;
    if (a.pad_mode < LERF_PAD_CONSTANT || a.pad_mode > LERF_PAD_WRAP) return LERF_EINVAL;
    int rc = launch_resize(a, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_warp(const lerf_plane_t* feat, const lerf_plane_t hyper[3], int H, int W, int C, const lerf_warp_geo_t* geo,
              int kind, double max_sigma, const lerf_mplane_t* out, void* stream) {
    if (!plane_ok(feat) || !geo || !out || !out->ptr || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    int nh = kind == LERF_KIND_GAUSS ? 3 : (kind == LERF_KIND_LINEAR ? 1 : 0);
    if (kind < LERF_KIND_GAUSS || kind > LERF_KIND_LANCZOS3) return LERF_EUNSUPPORTED;
    if (nh > 0 && !hyper) return LERF_EINVAL;
    for (int k = 0; k < nh; ++k)
        if (!hyper[k].ptr || hyper[k].dtype != hyper[0].dtype || hyper[k].sy != hyper[0].sy ||
            hyper[k].sx != hyper[0].sx || hyper[k].sc != hyper[0].sc)
            return LERF_EINVAL;
    if (geo->out_h < 1 || geo->out_w < 1) return LERF_EINVAL;
    WarpArgs a{};
    a.feat = feat->ptr; a.in_dtype = feat->dtype; a.fy = feat->sy; a.fx = feat->sx; a.fc = feat->sc;
    for (int k = 0; k < 3; ++k) a.h[k] = nh > 0 ? (k < nh ? hyper[k].ptr : hyper[0].ptr) : feat->ptr;
    a.h_dtype = nh > 0 ? hyper[0].dtype : feat->dtype;
    a.hy = nh > 0 ? hyper[0].sy : 0; a.hx = nh > 0 ? hyper[0].sx : 0; a.hc = nh > 0 ? hyper[0].sc : 0;
    a.H = H; a.W = W; a.C = C;
    a.geo.S = geo->S; a.geo.oH = geo->out_h; a.geo.oW = geo->out_w;
    memcpy(a.geo.minv, geo->minv, sizeof(a.geo.minv));
    a.geo.pad_r_lo = geo->pad_r_lo; a.geo.pad_r_hi = geo->pad_r_hi;
    a.geo.pad_c_lo = geo->pad_c_lo; a.geo.pad_c_hi = geo->pad_c_hi;
    a.geo.pad_mode = geo->pad_mode;
    a.geo.oy0 = geo->out_y0; a.geo.ox0 = geo->out_x0;
    if (geo->out_y0 < 0 || geo->out_x0 < 0 || geo->src_y0 < 0 || geo->src_y0 >= H) return LERF_EINVAL;
    if (geo->src_y0 > 0) {
        // the operands start at source row src_y0: addressed with the frame's row numbers through pointers moved back by that many rows
        // (never dereferenced outside the rows the caller passed: its guarantee, see lerf_warp_geo_t)
        auto back = [&](const void* p, int dtype, int64_t sy) {
            const int64_t es = dtype == LERF_U8 ? 1 : (dtype == LERF_F64 ? 8 : (dtype == LERF_I16 ? 2 : 4));
            return (const void*)((const char*)p - (int64_t)geo->src_y0 * sy * es);
        };
        a.feat = back(a.feat, a.in_dtype, a.fy);
        for (int k = 0; k < 3; ++k) a.h[k] = nh > 0 ? back(a.h[k], a.h_dtype, a.hy) : a.feat;
    }
    if (geo->pad_mode < LERF_PAD_CONSTANT || geo->pad_mode > LERF_PAD_WRAP) return LERF_EINVAL;
    if (geo->pad_mode != LERF_PAD_CONSTANT && out->dtype == LERF_U8) return LERF_EUNSUPPORTED;
    a.kind = kind; a.max_sigma = max_sigma;
    a.out = out->ptr; a.out_dtype = out->dtype; a.oy = out->sy; a.ox = out->sx; a.oc = out->sc;
    int rc = launch_warp(a, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_stages_packed_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C, const lerf_luts_t* luts,
                          uint32_t* packed, int64_t packed_sn, void* workspace, size_t workspace_bytes, void* stream) {
    if (!img || !luts || !packed || n < 1 || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    FusedArgs f{};
    f.img = img; f.in_sn = in_sn; f.n = n; f.H = H; f.W = W; f.C = C; f.luts = luts;
    f.S = 2; f.oH = H; f.oW = W; f.kind = luts->oC == 3 ? LERF_KIND_GAUSS : LERF_KIND_LINEAR;
    f.emit = packed; f.emit_sn = packed_sn; f.workspace = workspace; f.workspace_bytes = workspace_bytes;
    if (!fused_stages_supported(f)) return LERF_EUNSUPPORTED;
    if (workspace && workspace_bytes < fused_workspace_bytes(f)) return LERF_EINVAL;
    hipPointerAttribute_t pa;
    f.host_input = hipPointerGetAttributes(&pa, img) == hipSuccess && pa.type == hipMemoryTypeHost;
    int rc = launch_stages_fused(f, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

size_t lerf_stages_ragged_workspace_bytes(const lerf_stage_item_t* items, int n, int C) {
    if (!items || n < 1 || C < 1) return 0;
    size_t t = 0;
    for (int i = 0; i < n; ++i) t += ((size_t)items[i].H * items[i].W * C + 15) / 16 * 16;
    return t;
}

int lerf_stages_packed_ragged_u8(const lerf_stage_item_t* items, int n, int C, const lerf_luts_t* luts, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!items || !luts || n < 1 || n > 4096 || C < 1) return LERF_EINVAL;
    FusedItem its[64];
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int m = n - i0 < 64 ? n - i0 : 64;
        for (int i = 0; i < m; ++i) {
            const lerf_stage_item_t& s = items[i0 + i];
            if (!s.img || !s.packed || s.H < 1 || s.W < 1) return LERF_EINVAL;
            its[i] = FusedItem{s.img, nullptr, s.packed, s.H, s.W, s.H, s.W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        }
        FusedArgs f{};
        f.C = C; f.luts = luts; f.S = 2; f.kind = luts->oC == 3 ? LERF_KIND_GAUSS : LERF_KIND_LINEAR;
        f.items = its; f.n_items = m; f.workspace = workspace; f.workspace_bytes = workspace_bytes;
        if (!fused_stages_supported(f)) return LERF_EUNSUPPORTED;
        if (workspace && workspace_bytes < fused_workspace_bytes(f)) return LERF_EINVAL;
        int rc = launch_stages_fused(f, as_stream(stream));
        if (rc != LERF_OK) return rc;
    }
    return check_launch();
}

int lerf_warp_tile_boxes(const lerf_warp_geo_t* geo, int H, int W, int32_t* boxes) {
    if (!geo || geo->S != 2) return LERF_EINVAL;
    return host::warp_tile_boxes(geo->minv, geo->pad_r_lo, geo->pad_c_lo, H, W, geo->out_h, geo->out_w, 64, boxes);
}

static void warp_fused_args(FusedArgs& f, WarpGeo& wg, int n, int H, int W, int C, const lerf_luts_t* luts, const lerf_warp_geo_t* geo, int kind,
                            double max_sigma) {
    wg.S = geo->S; wg.oH = geo->out_h; wg.oW = geo->out_w;
    memcpy(wg.minv, geo->minv, sizeof(wg.minv));
    wg.pad_r_lo = geo->pad_r_lo; wg.pad_r_hi = geo->pad_r_hi; wg.pad_c_lo = geo->pad_c_lo; wg.pad_c_hi = geo->pad_c_hi;
    wg.pad_mode = geo->pad_mode;
    wg.oy0 = geo->out_y0; wg.ox0 = geo->out_x0;
    f.n = n; f.H = H; f.W = W; f.C = C; f.luts = luts; f.S = geo->S; f.oH = geo->out_h; f.oW = geo->out_w;
    f.kind = kind; f.max_sigma = (float)max_sigma; f.wgeo = &wg;
}

int lerf_warp_fused_supported(int C, const lerf_luts_t* luts, const lerf_warp_geo_t* geo, int H, int W, int kind, double max_sigma) {
    if (!luts || !geo || H < 1 || W < 1 || geo->out_y0 != 0 || geo->out_x0 != 0 || geo->src_y0 != 0) return 0;
    FusedArgs f{};
    WarpGeo wg{};
    warp_fused_args(f, wg, 1, H, W, C, luts, geo, kind, max_sigma);
    return warp_fused_supported(f) ? 1 : 0;
}

int lerf_warp_fused_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C, const lerf_luts_t* luts, const lerf_warp_geo_t* geo,
                       const int32_t* tile_boxes, int kind, double max_sigma, uint8_t* out, int64_t out_sn, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!img || !luts || !geo || !tile_boxes || !out || !workspace || n < 1 || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    if (workspace_bytes < lerf_sr_fused_workspace_bytes(H, W, C, n)) return LERF_EINVAL;
    if (geo->out_y0 != 0 || geo->out_x0 != 0 || geo->src_y0 != 0) return LERF_EUNSUPPORTED;
    FusedArgs f{};
    WarpGeo wg{};
    warp_fused_args(f, wg, n, H, W, C, luts, geo, kind, max_sigma);
    f.img = img; f.in_sn = in_sn; f.out = out; f.out_sn = out_sn; f.workspace = workspace; f.workspace_bytes = workspace_bytes;
    f.wboxes = tile_boxes;
    int rc = launch_warp_fused(f, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_unpack_stages(const uint32_t* packed, int64_t n_pxch, int oC, uint8_t* feat, uint8_t* hq, void* stream) {
    if (!packed || n_pxch < 1 || (!feat && !hq)) return LERF_EINVAL;
    int rc = launch_unpack_stages(packed, n_pxch, oC, feat, hq, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_warp_packed(const uint32_t* packed, int64_t packed_sn, int n, int H, int W, int C, const lerf_warp_geo_t* geo, int kind,
                     double max_sigma, const lerf_mplane_t* out, int64_t out_sn, void* stream) {
    if (!packed || !geo || !out || !out->ptr || n < 1 || H < 1 || W < 1 || C < 1 || geo->out_h < 1 || geo->out_w < 1)
        return LERF_EINVAL;
    WarpGeo g;
    g.S = geo->S; g.oH = geo->out_h; g.oW = geo->out_w;
    memcpy(g.minv, geo->minv, sizeof(g.minv));
    g.pad_r_lo = geo->pad_r_lo; g.pad_r_hi = geo->pad_r_hi; g.pad_c_lo = geo->pad_c_lo; g.pad_c_hi = geo->pad_c_hi;
    g.pad_mode = LERF_PAD_CONSTANT;
    g.oy0 = geo->out_y0; g.ox0 = geo->out_x0;
    if (geo->pad_mode != LERF_PAD_CONSTANT) return LERF_EUNSUPPORTED;
    if (geo->out_y0 < 0 || geo->out_x0 < 0 || geo->src_y0 < 0 || geo->src_y0 >= H) return LERF_EINVAL;
    packed -= (int64_t)geo->src_y0 * W * C;                    // rows of the frame from src_y0 on (lerf_warp_geo_t)
    int rc = launch_warp_packed(packed, packed_sn, n, H, W, C, g, kind, (float)max_sigma, out->ptr, out->dtype, out->sy, out->sx,
                                out->sc, out_sn, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_rect_copy_u8(uint8_t* frames, int n, int fh, int fw, int C, uint8_t* staging, const lerf_rect_t* rects, int n_rects,
                      int to_staging, void* stream) {
    if (!frames || !staging || !rects || n < 1 || fh < 1 || fw < 1 || C < 1 || n_rects < 1 || n_rects > LERF_MAX_RECTS) return LERF_EINVAL;
    for (int r = 0; r < n_rects; ++r)
        if (rects[r].y < 0 || rects[r].x < 0 || rects[r].h < 1 || rects[r].w < 1 || rects[r].y + rects[r].h > fh ||
            rects[r].x + rects[r].w > fw || rects[r].off < 0)
            return LERF_EINVAL;
    int rc = launch_rect_copy(frames, n, fh, fw, C, staging, rects, n_rects, to_staging, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

size_t lerf_srnet_weight_floats(int outC) { return outC >= 1 && outC <= 4 ? srnet_weight_floats(outC) : 0; }

int lerf_srnet_to_lut(const float* weights, int outC, int interval, int8_t* lut, float* y, void* stream) {
    if (!weights || !lut) return LERF_EINVAL;
    int rc = launch_srnet_to_lut(weights, outC, interval, lut, y, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

int lerf_ubench_lds_gather(int pattern, int iters, int workgroups, uint32_t* sink, void* stream) {
    if ((pattern != 0 && pattern != 1) || iters < 1 || workgroups < 1 || !sink) return LERF_EINVAL;
    int rc = launch_ubench_lds_gather(pattern, iters, workgroups, sink, as_stream(stream));
    return rc != LERF_OK ? rc : check_launch();
}

static size_t general_workspace_bytes(int H, int W, int C, int n) { return (size_t)n * H * W * C * 4; }

size_t lerf_sr_fused_workspace_bytes(int H, int W, int C, int n) {
    // Two uses, the larger one sizes it: (a) the two-launch tile-fused path parks the stage-1 output of the batch there
    // (n frames, frame stride rounded up to 16 bytes) between s1_kernel and the stage-2/3 launch; (b) the general
    // fallback keeps feat + up to 3 hyper planes per frame.
    if (H < 1 || W < 1 || C < 1 || n < 1) return 0;
    const size_t hwc = (size_t)H * W * C;
    const size_t a = (size_t)n * ((hwc + 15) / 16 * 16) + 512;                     // (a) + alignment slack
    const size_t b = general_workspace_bytes(H, W, C, n);
    return a > b ? a : b;
}

static void fused_args_of(FusedArgs& f, int C, const lerf_luts_t* luts, const lerf_sr_geo_t* geo, int H, int W, int kind) {
    f.H = H; f.W = W; f.C = C; f.luts = luts;
    f.S = geo->S; f.oH = geo->out_h; f.oW = geo->out_w;
    f.left_r = geo->left_r; f.dis_r = geo->dis_r; f.left_c = geo->left_c; f.dis_c = geo->dis_c;
    f.dis_r64 = (geo->dis_r64 && geo->dis_c64) ? geo->dis_r64 : nullptr;      // both or neither: tie guard
    f.dis_c64 = (geo->dis_r64 && geo->dis_c64) ? geo->dis_c64 : nullptr;
    f.kind = kind;
    f.roi_y = geo->roi_y; f.roi_x = geo->roi_x; f.roi_h = geo->roi_h; f.roi_w = geo->roi_w;
    f.tq_cap = geo->tie_queue_cap; f.pad_mode = geo->pad_mode; f.flags = geo->flags; f.out_pitch = geo->out_row_pitch;
}

int lerf_sr_fused_supported(int C, const lerf_luts_t* luts, const lerf_sr_geo_t* geo, int H, int W, int kind, double max_sigma) {
    if (!luts || !geo || H < 1 || W < 1 || C < 1) return 0;
    if (kind != LERF_KIND_GAUSS && kind != LERF_KIND_LINEAR) return 0;
    if ((kind == LERF_KIND_GAUSS) != (luts->oC == 3)) return 0;
    FusedArgs f{};
    fused_args_of(f, C, luts, geo, H, W, kind);
    f.n = 1;
    f.max_sigma = (float)max_sigma;
    f.workspace = (void*)1;          // "a workspace will be passed" (wrap padding needs the two-launch path)
    return fused_supported(f) ? 1 : 0;
}

int lerf_sr_fused_u8(const uint8_t* img, int64_t in_sn, int n, int H, int W, int C, const lerf_luts_t* luts,
                     const lerf_sr_geo_t* geo, int kind, double max_sigma, uint8_t* out, int64_t out_sn,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!img || !out || !luts || !geo || n < 1 || H < 1 || W < 1 || C < 1) return LERF_EINVAL;
    if (!geo->left_r || !geo->left_c || !geo->dis_r || !geo->dis_c) return LERF_EINVAL;
    if (kind != LERF_KIND_GAUSS && kind != LERF_KIND_LINEAR) return LERF_EUNSUPPORTED;
    if ((kind == LERF_KIND_GAUSS) != (luts->oC == 3)) return LERF_EINVAL;
    if (geo->pad_mode < LERF_PAD_CONSTANT || geo->pad_mode > LERF_PAD_WRAP) return LERF_EINVAL;
    if (geo->out_row_pitch != 0 && (int64_t)geo->out_row_pitch < (int64_t)geo->out_w * C) return LERF_EINVAL;
    const bool roi = geo->roi_h > 0 && geo->roi_w > 0;
    if (roi && (geo->roi_y < 0 || geo->roi_x < 0 || geo->roi_y + geo->roi_h > H || geo->roi_x + geo->roi_w > W)) return LERF_EINVAL;
    FusedArgs f{};
    fused_args_of(f, C, luts, geo, H, W, kind);
    f.img = img; f.in_sn = in_sn; f.n = n;
    f.max_sigma = (float)max_sigma; f.out = out; f.out_sn = out_sn; f.workspace = workspace; f.workspace_bytes = workspace_bytes;
    if (fused_supported(f)) {
        if (workspace && workspace_bytes < fused_workspace_bytes(f)) return LERF_EINVAL;
        // frames in pinned host memory (read over PCIe from inside the kernel, lerf-pytorch_amd/stream.py) are fetched once per
        // tile into LDS; device frames let stage 1 read its neighbourhood pixels over the vector-memory path (L1 / L2 hits)
        // (LERF_GEO_INPUT_DEVICE / _HOST: the caller knows where its frames live and saves the runtime query per call)
        if (geo->flags & (LERF_GEO_INPUT_DEVICE | LERF_GEO_INPUT_HOST)) {
            f.host_input = (geo->flags & LERF_GEO_INPUT_HOST) != 0;
        } else {
            hipPointerAttribute_t pa;
            f.host_input = hipPointerGetAttributes(&pa, img) == hipSuccess && pa.type == hipMemoryTypeHost;
        }
        int rc = launch_sr_fused(f, as_stream(stream));
        return rc != LERF_OK ? rc : check_launch();
    }
    // general configuration: the three stages through the caller's workspace
    if (!workspace || roi) return roi ? LERF_EUNSUPPORTED : LERF_EINVAL;
    if (workspace_bytes < general_workspace_bytes(H, W, C, n)) return LERF_EINVAL;
    const int oC = luts->oC;
    for (int b = 0; b < n; ++b) {
        uint8_t* ws = (uint8_t*)workspace + (size_t)b * H * W * C * 4;
        lerf_plane_t in{img + b * in_sn, LERF_U8, (int64_t)W * C, C, 1};
        lerf_mplane_t feat{ws, LERF_U8, (int64_t)W * C, C, 1};
        lerf_mplane_t hyp{ws + (size_t)H * W * C, LERF_U8, (int64_t)W * C * oC, (int64_t)C * oC, oC};
        int rc = lerf_lut_stages_u8(&in, H, W, C, luts, &feat, &hyp, stream);
        if (rc != LERF_OK) return rc;
        lerf_plane_t fin{ws, LERF_U8, (int64_t)W * C, C, 1};
        lerf_plane_t hp[3];
        for (int k = 0; k < 3; ++k)
            hp[k] = lerf_plane_t{ws + (size_t)H * W * C + (k < oC ? k : 0), LERF_U8, (int64_t)W * C * oC,
                                 (int64_t)C * oC, oC};
        lerf_mplane_t o{out + b * out_sn, LERF_U8, geo->out_row_pitch ? (int64_t)geo->out_row_pitch : (int64_t)geo->out_w * C, C, 1};
        rc = lerf_resize(&fin, hp, H, W, C, geo, kind, max_sigma, &o, stream);
        if (rc != LERF_OK) return rc;
    }
    return LERF_OK;
}

size_t lerf_sr_ragged_workspace_bytes(const lerf_sr_item_t* items, int n, int C) {
    if (!items || n < 1 || C < 1) return 0;
    size_t sum = 0, mx = 0;
    for (int i = 0; i < n; ++i) {
        sum += ((size_t)items[i].H * items[i].W * C + 15) / 16 * 16;
        const size_t g = lerf_sr_fused_workspace_bytes(items[i].H, items[i].W, C, 1);
        mx = g > mx ? g : mx;
    }
    return sum > mx ? sum : mx;
}

int lerf_sr_fused_ragged_u8(const lerf_sr_item_t* items, int n, int C, const lerf_luts_t* luts, int kind, double max_sigma,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!items || !luts || n < 1 || n > 4096 || C < 1 || !workspace) return LERF_EINVAL;
    if (kind != LERF_KIND_GAUSS && kind != LERF_KIND_LINEAR) return LERF_EUNSUPPORTED;
    if ((kind == LERF_KIND_GAUSS) != (luts->oC == 3)) return LERF_EINVAL;
    if (workspace_bytes < lerf_sr_ragged_workspace_bytes(items, n, C)) return LERF_EINVAL;
    bool all = true;
    for (int i = 0; i < n; ++i) {
        const lerf_sr_item_t& s = items[i];
        if (!s.img || !s.out || s.H < 1 || s.W < 1 || !s.geo.left_r || !s.geo.left_c || !s.geo.dis_r || !s.geo.dis_c) return LERF_EINVAL;
        if (s.geo.S != items[0].geo.S || s.geo.pad_mode != items[0].geo.pad_mode) return LERF_EINVAL;
        // per-call knobs of the launch: one value for all items (the first item's would silently win otherwise)
        // (LERF_GEO_X2_TABLES / _INPUT_* describe an item; FORCE_GENERAL / SINGLE_LAUNCH steer the launch)
        if (s.geo.tie_queue_cap != items[0].geo.tie_queue_cap || s.geo.out_row_pitch != 0 ||
            ((s.geo.flags ^ items[0].geo.flags) & (LERF_GEO_FORCE_GENERAL | LERF_GEO_SINGLE_LAUNCH)) != 0)
            return LERF_EINVAL;
        // (the float64 tables -- the tie guard -- are honoured frame by frame: FrameDesc.dis_r64)
        all = all && lerf_sr_fused_supported(C, luts, &s.geo, s.H, s.W, kind, max_sigma) && s.geo.roi_h == 0;
    }
    if (!all) {                       // some item has no tile-fused kernel: item by item (same results)
        for (int i = 0; i < n; ++i) {
            const lerf_sr_item_t& s = items[i];
            int rc = lerf_sr_fused_u8(s.img, 0, 1, s.H, s.W, C, luts, &s.geo, kind, max_sigma, s.out, 0, workspace, workspace_bytes, stream);
            if (rc != LERF_OK) return rc;
        }
        return LERF_OK;
    }
    FusedItem its[64];
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int m = n - i0 < 64 ? n - i0 : 64;
        for (int i = 0; i < m; ++i) {
            const lerf_sr_item_t& s = items[i0 + i];
            its[i] = FusedItem{s.img, s.out, nullptr, s.H, s.W, s.geo.out_h, s.geo.out_w, s.geo.left_r, s.geo.dis_r,
                               s.geo.left_c, s.geo.dis_c, s.geo.dis_r64, s.geo.dis_c64};
        }
        FusedArgs f{};
        fused_args_of(f, C, luts, &items[0].geo, items[i0].H, items[i0].W, kind);
        f.roi_h = f.roi_w = 0;
        f.items = its; f.n_items = m; f.max_sigma = (float)max_sigma; f.workspace = workspace; f.workspace_bytes = workspace_bytes;
        int rc = launch_sr_fused(f, as_stream(stream));
        if (rc != LERF_OK) return rc;
    }
    return check_launch();
}

}  // extern "C"
