"""Positional encoding, same surface as the reference's models/embedder.py:39-51
(`get_embedder(multires, input_dims) -> (embed_fn, out_dim)`); the arithmetic runs in the
`nudf_posenc` HIP kernel.  Inside the networks the encoding is written straight into the
first GEMM's A-operand buffer (see mlp.py), so this standalone function is only used by
callers that want the encoding itself."""
import torch

from .._lib import call, ptr


class Embedder:
    def __init__(self, input_dims, num_freqs):
        self.input_dims = input_dims
        self.num_freqs = num_freqs
        self.out_dim = input_dims * (2 * num_freqs + 1)

    def embed(self, inputs):
        x = inputs.detach().contiguous().float()
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.input_dims)
        out = torch.empty(x2.shape[0], self.out_dim, device=x.device)
        call("nudf_posenc", ptr(x2), self.input_dims, 1, None, self.input_dims, self.num_freqs, 1.0, x2.shape[0],
             ptr(out), self.out_dim, 1.0, None, 0, 0.0)
        return out.reshape(*lead, self.out_dim)


def get_embedder(multires, input_dims=3):
    eo = Embedder(input_dims, multires)

    def embed(x, eo=eo):
        return eo.embed(x)

    return embed, eo.out_dim
