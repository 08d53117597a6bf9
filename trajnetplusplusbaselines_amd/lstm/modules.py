"""Parameter containers with the class names, constructor arguments and state_dict keys of the reference's
lstm/modules.py, so that checkpoints load unchanged.  Inside ``LSTM.forward`` both layers are fused into the per-step
kernels (csrc/lstm_seq.hip: ``track_prepare_kernel``); the stand-alone forwards below exist for API parity and run
their Linear on the fp32 MFMA GEMM."""
import torch

from .. import _lib

_N_TAGS = 2   # the two trailing "start of encoder / start of decoder" tag columns of an embedding


class InputEmbedding(torch.nn.Module):
    """ReLU(Linear(scale * velocity)) followed by the tag columns (reference lstm/modules.py:4-48).  The reference
    scales the input because ReLU units with the default bias initialisation would otherwise stay inactive."""

    def __init__(self, input_dim, embedding_dim, scale, use_tags=True):
        super().__init__()
        self.embedding_dim, self.scale, self.use_tags = embedding_dim, scale, use_tags
        width = embedding_dim - _N_TAGS if use_tags else embedding_dim
        # Sequential(Linear, ReLU) keeps the checkpoint key "input_embeddings.0.{weight,bias}"
        self.input_embeddings = torch.nn.Sequential(torch.nn.Linear(input_dim, width), torch.nn.ReLU())

    def forward(self, vel):
        layer = self.input_embeddings[0]
        _lib.require_device(layer.weight, 'InputEmbedding parameters')
        x = _lib.f32c(vel, layer.weight.device) * self.scale
        out = x.new_zeros(x.size(0), self.embedding_dim)          # tag columns stay zero
        _lib.linear_forward(x, layer.weight.detach(), layer.bias.detach(), relu=True, out=out[:, :layer.out_features])
        return out

    def _tag(self, vel, column):
        if not self.use_tags:
            raise Exception('Input embedding does not support start tag')
        tag = torch.zeros(vel.size(0), self.embedding_dim, device=vel.device)
        tag[:, column] = 1
        return tag

    def start_enc(self, vel):
        """one-hot "encoder start" row per track (reference lstm/modules.py:32-39)"""
        return self._tag(vel, -2)

    def start_dec(self, vel):
        """one-hot "decoder start" row per track (reference lstm/modules.py:41-48)"""
        return self._tag(vel, -1)


class Hidden2Normal(torch.nn.Module):
    """hidden state -> (mu_x, mu_y, sigma_x, sigma_y, rho) with sigma in (0.01, 0.21) and rho in (0, 0.7)
    (reference lstm/modules.py:51-64)."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.linear = torch.nn.Linear(hidden_dim, 5)

    def forward(self, hidden_state):
        _lib.require_device(self.linear.weight, 'Hidden2Normal parameters')
        raw = _lib.linear_forward(hidden_state, self.linear.weight.detach(), self.linear.bias.detach())
        squashed = torch.sigmoid(raw[:, 2:])
        return torch.cat([raw[:, :2], 0.01 + 0.2 * squashed[:, :2], 0.7 * squashed[:, 2:]], dim=1)
