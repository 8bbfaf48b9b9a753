// f32 instantiation of the RenderingNetwork backward kernel - see k_render_bwd.h
#include "k_render_bwd.h"
extern "C" int vdn_rendernet_bwd_f32(const VdnRenderNetBwdArgs* args, void* stream) { return vdn::launch_rendernet_bwd<vdn::F32>(args, stream); }
