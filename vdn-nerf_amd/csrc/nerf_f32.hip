// f32 instantiation of the background NeRF forward kernel - see k_nerf_fwd.h
#include "k_nerf_fwd.h"
extern "C" int vdn_nerf_mlp_fwd_f32(const VdnNerfArgs* args, void* stream) { return vdn::launch_nerf_fwd<vdn::F32>(args, stream); }
