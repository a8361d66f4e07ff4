// General tile-fused kernels (any 1..4 sampling patterns per stage, scale factors up to 8, ragged launches) for 1-channel
// frames: one instance of lerf_fused_impl.h.  Dispatch: lerf_fused.hip.
#define LERF_FUSED_NS fused_c1
#define LERF_FUSED_CH 1
#include "lerf_fused_impl.h"

namespace lerf {
int launch_sr_fused_c1(const FusedArgs& a, hipStream_t st) { return fused_c1::launch_sr<true>(a, st); }
int launch_stages_fused_c1(const FusedArgs& a, hipStream_t st) { return fused_c1::launch_stages<true>(a, st); }
}  // namespace lerf
