// bf16 instantiation of the RenderingNetwork forward kernel - see k_render_fwd.h
#include "k_render_fwd.h"
extern "C" int vdn_rendernet_fwd_bf16(const VdnRenderNetArgs* args, void* stream) { return vdn::launch_rendernet_fwd<vdn::BF16>(args, stream); }
