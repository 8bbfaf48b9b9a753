// f32 instantiation of the SDF backward kernels (rbar / fbar) - see k_sdf_bwd.h
#include "k_sdf_bwd.h"
extern "C" int vdn_sdf_bwd_rbar_f32(const VdnSdfRbarArgs* args, void* stream) { return vdn::launch_sdf_rbar<vdn::F32>(args, stream); }
extern "C" int vdn_sdf_bwd_fbar_f32(const VdnSdfFbarArgs* args, void* stream) { return vdn::launch_sdf_fbar<vdn::F32>(args, stream); }
