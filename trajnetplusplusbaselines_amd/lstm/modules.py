"""Parameter containers mirroring the reference's lstm/modules.py (same class names, state_dict keys and
initialisation), with forwards that run on the MI355X kernels.  Inside ``LSTM.forward`` these layers are
fused into the per-step kernels (csrc/lstm_seq.hip); the standalone forwards below exist for API parity."""
import torch

from .. import _lib


class InputEmbedding(torch.nn.Module):
    """Linear embedding, ReLU non-linearity, input scaling (reference lstm/modules.py:4-48)."""

    def __init__(self, input_dim, embedding_dim, scale, use_tags=True):
        super(InputEmbedding, self).__init__()
        self.embedding_dim = embedding_dim
        self.scale = scale
        self.use_tags = use_tags
        linear_embedding_dim = self.embedding_dim - (2 if use_tags else 0)
        self.input_embeddings = torch.nn.Sequential(
            torch.nn.Linear(input_dim, linear_embedding_dim),
            torch.nn.ReLU(),
        )

    def forward(self, vel):
        lin = self.input_embeddings[0]
        _lib.require_device(lin.weight, 'InputEmbedding parameters')
        vel = _lib.f32c(vel, lin.weight.device)
        out = torch.zeros(vel.size(0), self.embedding_dim, dtype=torch.float32, device=vel.device)
        n_lin = lin.weight.shape[0]
        _lib.linear_forward(vel * self.scale, lin.weight.detach(), lin.bias.detach(), relu=True, out=out[:, :n_lin])
        return out

    def start_enc(self, vel):
        """Start tag (reference lstm/modules.py:32-39)."""
        if not self.use_tags:
            raise Exception('Input embedding does not support start tag')
        v = torch.zeros(vel.size(0), self.embedding_dim, device=vel.device)
        v[:, -2] = 1
        return v

    def start_dec(self, vel):
        """Start tag (reference lstm/modules.py:41-48)."""
        if not self.use_tags:
            raise Exception('Input embedding does not support start tag')
        v = torch.zeros(vel.size(0), self.embedding_dim, device=vel.device)
        v[:, -1] = 1
        return v


class Hidden2Normal(torch.nn.Module):
    """Linear(H -> 5) with the sigma / rho output ranges of reference lstm/modules.py:51-64."""

    def __init__(self, hidden_dim):
        super(Hidden2Normal, self).__init__()
        self.linear = torch.nn.Linear(hidden_dim, 5)

    def forward(self, hidden_state):
        _lib.require_device(self.linear.weight, 'Hidden2Normal parameters')
        normal = _lib.linear_forward(hidden_state, self.linear.weight.detach(), self.linear.bias.detach())
        normal[:, 2] = 0.01 + 0.2 * torch.sigmoid(normal[:, 2])
        normal[:, 3] = 0.01 + 0.2 * torch.sigmoid(normal[:, 3])
        normal[:, 4] = 0.7 * torch.sigmoid(normal[:, 4])
        return normal
