// fp32 (parity) instantiation of the SDF forward kernel - see k_sdf_fwd.h
#include "k_sdf_fwd.h"
extern "C" int vdn_sdf_mlp_fwd_f32(int mode, const VdnSdfArgs* args, void* stream) {
    return vdn::launch_sdf_fwd<vdn::F32>(mode, args, stream);
}
