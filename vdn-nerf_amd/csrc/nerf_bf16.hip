// bf16 background NeRF forward: the flat-stream kernel - see k_nerf_fwd2.h
#include "k_nerf_fwd2.h"
extern "C" int vdn_nerf_mlp_fwd_bf16(const VdnNerfArgs* args, void* stream) { return vdn::launch_nerf_fwd2(args, stream); }
