// Tile-fused uint8 SR path (stages 1+2+3 in one launch).  Placeholder until the
// tiled kernel lands: reports "not supported" so the API takes the 3-launch path.
#include "lerf_kernels.h"

namespace lerf {
bool fused_supported(const FusedArgs&) { return false; }
int launch_sr_fused(const FusedArgs&, hipStream_t) { return LERF_EUNSUPPORTED; }
}  // namespace lerf
