// Layer-wise SDF engine (k_sdf_lw.h) - its own translation unit (same flags as sdf_bf16.hip, vdn_hip/build.py).
#include "k_sdf_lw.h"
namespace vdn {
int sdf_lw0_launch(const VdnSdfArgs* args, hipStream_t stream) { return sdflw::launch0<>(args, stream); }
}  // namespace vdn
