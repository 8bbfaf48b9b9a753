// bf16 instantiation of the background NeRF forward kernel - see k_nerf_fwd.h
#include "k_nerf_fwd.h"
extern "C" int vdn_nerf_mlp_fwd_bf16(const VdnNerfArgs* args, void* stream) { return vdn::launch_nerf_fwd<vdn::BF16>(args, stream); }
