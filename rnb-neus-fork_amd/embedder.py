"""Positional encoding with the reference's interface (models/embedder.py:58-74):
`embed_fn, out_dim = get_embedder(multires, input_dims)`.

Inside the renderer the encoding is computed by the HIP kernels (csrc/mlp.hip: pe_points_kernel,
color_input_kernel); this torch version exists for API compatibility of code that calls the closure
directly and defines the column order gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., cos(2^{L-1} x)]."""
import torch


class Embedder:
    def __init__(self, input_dims, multires):
        self.input_dims = input_dims
        self.multires = multires
        self.out_dim = input_dims * (1 + 2 * multires)

    def embed(self, x):
        cols = [x]
        for k in range(self.multires):
            f = float(2 ** k)
            cols.append(torch.sin(x * f))
            cols.append(torch.cos(x * f))
        return torch.cat(cols, dim=-1)


def get_embedder(multires, input_dims=3):
    eo = Embedder(input_dims, multires)
    return eo.embed, eo.out_dim
