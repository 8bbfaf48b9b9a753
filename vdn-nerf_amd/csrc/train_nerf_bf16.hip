// bf16 instantiation of the background NeRF backward kernel - see k_nerf_bwd.h
#include "k_nerf_bwd.h"
extern "C" int vdn_nerf_mlp_bwd_bf16(const VdnNerfBwdArgs* args, void* stream) { return vdn::launch_nerf_bwd<vdn::BF16>(args, stream); }
