// f32 instantiation of the RenderingNetwork forward kernel - see k_render_fwd.h
#include "k_render_fwd.h"
extern "C" int vdn_rendernet_fwd_f32(const VdnRenderNetArgs* args, void* stream) { return vdn::launch_rendernet_fwd<vdn::F32>(args, stream); }
