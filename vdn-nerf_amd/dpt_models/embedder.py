"""Positional encoding (drop-in for the reference's dpt_models/embedder.py).

Inside the networks the sin/cos octaves are evaluated in registers by the fused MLP kernels (csrc/vdn_common.h:
posenc), so the networks never call this module's embed(). It is kept callable for users of the reference's
`embed_fn` attributes (e.g. sdf_network.embed_fn_fine): get_embedder(multires, input_dims) -> (embed_fn, out_dim),
channel order [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...] (embedder.py:27-36), evaluated by the
vdn_posenc kernel of libvdn_render.so. No CPU path.
"""
import ctypes

import torch

from vdn_hip import lib


class Embedder:
    def __init__(self, input_dims, num_freqs, include_input=True):
        if not include_input:
            raise ValueError("include_input=False is not used by any shipped configuration")
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.out_dim = input_dims * (1 + 2 * num_freqs)

    def embed(self, inputs):
        if not (torch.is_tensor(inputs) and inputs.is_cuda and inputs.dtype == torch.float32):
            raise RuntimeError("Embedder.embed: expected a float32 tensor on the MI355X (cuda) device; this package has no CPU path")
        if inputs.shape[-1] != self.input_dims:
            raise ValueError("Embedder.embed: last dimension is %d, expected %d" % (inputs.shape[-1], self.input_dims))
        x = inputs.detach().reshape(-1, self.input_dims).contiguous()
        out = torch.empty(x.shape[0], self.out_dim, dtype=torch.float32, device=x.device)
        if x.shape[0] > 0:
            lib.call("vdn_posenc", lib.ptr(x), lib.ptr(out), ctypes.c_int64(x.shape[0]), self.input_dims, self.num_freqs,
                     torch.cuda.current_stream().cuda_stream)
        return out.reshape(*inputs.shape[:-1], self.out_dim)


def get_embedder(multires, input_dims=3):
    eo = Embedder(input_dims, multires)
    return eo.embed, eo.out_dim
