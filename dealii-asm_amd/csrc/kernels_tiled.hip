// Tiled owner-computes sweeps (placeholder until the kernels land): returning
// false makes the context report ASM_ERR_NOT_SUPPORTED, it never falls back.
#include "kernels.h"

namespace asmamd
{
  bool
  tiled_kernels_available()
  {
    return false;
  }
  bool
  laplace_tiled(const DevGrid &, const DevOp &, const TiledArgs &, hipStream_t)
  {
    return false;
  }
  bool
  fdm_tiled(const DevGrid &, const DevFdm &, const TiledArgs &, hipStream_t)
  {
    return false;
  }
} // namespace asmamd
