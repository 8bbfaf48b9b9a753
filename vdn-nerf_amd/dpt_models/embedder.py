"""Positional encoding metadata (drop-in for the reference's dpt_models/embedder.py).

On the MI355X path the sin/cos octaves are evaluated inside the fused MLP kernels
(csrc/vdn_common.h: posenc), so this module only carries the shape contract of
embedder.py:39-51: get_embedder(multires, input_dims) -> (embed_fn, out_dim), channel order
[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...].
"""


class Embedder:
    def __init__(self, input_dims, num_freqs, include_input=True):
        if not include_input:
            raise ValueError("include_input=False is not used by any shipped configuration")
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.out_dim = input_dims * (1 + 2 * num_freqs)

    def embed(self, inputs):
        raise RuntimeError("positional encoding is fused into the HIP MLP kernels; there is no "
                           "standalone (eager) embed on this path")


def get_embedder(multires, input_dims=3):
    eo = Embedder(input_dims, multires)
    return eo.embed, eo.out_dim
