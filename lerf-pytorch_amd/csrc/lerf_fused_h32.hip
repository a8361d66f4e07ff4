// General tile-fused kernels for 3-channel frames with 32-row tiles: one more instance of lerf_fused_impl.h, taken by launches
// that cannot fill the chip with 64-row tiles (lerf_fused.hip: tile_rows_for).  A tile's fixed costs (18 piece copies, LUT loads,
// barriers) do not shrink with it, so these tiles cost more per pixel -- they buy latency, not throughput.
#define LERF_FUSED_NS fused_h32
#define LERF_FUSED_CH 3
#define LERF_FUSED_TH 32
#include "lerf_fused_impl.h"

namespace lerf {
int launch_sr_fused_h32(const FusedArgs& a, hipStream_t st) { return fused_h32::launch_sr<true>(a, st); }
int launch_stages_fused_h32(const FusedArgs& a, hipStream_t st) { return fused_h32::launch_stages<true>(a, st); }
}  // namespace lerf
