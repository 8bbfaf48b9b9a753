// bf16-MFMA (throughput) instantiation of the SDF forward kernel - see k_sdf_fwd.h
#include "k_sdf_fwd.h"
extern "C" int vdn_sdf_mlp_fwd_bf16(int mode, const VdnSdfArgs* args, void* stream) {
    return vdn::launch_sdf_fwd<vdn::BF16>(mode, args, stream);
}
