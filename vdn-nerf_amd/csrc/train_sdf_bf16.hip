// bf16 instantiation of the SDF backward kernels (rbar / fbar) - see k_sdf_bwd.h
#include "k_sdf_bwd.h"
extern "C" int vdn_sdf_bwd_rbar_bf16(const VdnSdfRbarArgs* args, void* stream) { return vdn::launch_sdf_rbar<vdn::BF16>(args, stream); }
extern "C" int vdn_sdf_bwd_fbar_bf16(const VdnSdfFbarArgs* args, void* stream) { return vdn::launch_sdf_fbar<vdn::BF16>(args, stream); }
