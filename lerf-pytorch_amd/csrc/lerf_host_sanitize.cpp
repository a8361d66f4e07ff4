// The host-only entry points of include/lerf_hip.h compiled WITHOUT HIP, for the AddressSanitizer + UndefinedBehaviorSanitizer
// build of the CPU suite (`make -C lerf-pytorch_amd/csrc asan` -> build_asan/liblerf_host_asan.so; GPU sanitizers are not
// available on this pool).  Same code as liblerf_hip.so: both include lerf_host_geometry.h.
#include "lerf_host_geometry.h"

using namespace lerf;

extern "C" {
int lerf_abi_version(void) { return LERF_ABI_VERSION; }
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
int lerf_warp_tile_boxes(const lerf_warp_geo_t* geo, int H, int W, int32_t* boxes) {
    if (!geo || geo->S != 2) return LERF_EINVAL;
    return host::warp_tile_boxes(geo->minv, geo->pad_r_lo, geo->pad_c_lo, H, W, geo->out_h, geo->out_w, 64, boxes);
}
}
