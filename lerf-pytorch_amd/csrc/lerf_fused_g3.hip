// General tile-fused kernels (any 1..4 sampling patterns per stage, scale factors up to 8, ragged launches) for 3-channel
// frames: one instance of lerf_fused_impl.h.  Dispatch: lerf_fused.hip.
#define LERF_FUSED_NS fused_g3
#define LERF_FUSED_CH 3
#include "lerf_fused_impl.h"

namespace lerf {
int launch_sr_fused_g3(const FusedArgs& a, hipStream_t st) { return fused_g3::launch_sr<true>(a, st); }
int launch_stages_fused_g3(const FusedArgs& a, hipStream_t st) { return fused_g3::launch_stages<true>(a, st); }
}  // namespace lerf
