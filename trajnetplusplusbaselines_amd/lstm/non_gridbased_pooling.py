"""Non-grid interaction modules on the MI355X path (reference lstm/non_gridbased_pooling.py).

Same constructor arguments, sub-module names (=> state_dict keys) and the same
``forward(hidden_state [B,N,H], obs1 [B,N,2], obs2 [B,N,2]) -> [B*N, out_dim]`` contract as the reference; the all-pairs
work runs in csrc/pool_nongrid.hip.  Inside ``LSTM.forward`` the modules are driven by the fused sequence kernel
(pool types TNP_POOL_NN / TNP_POOL_HIDDENMLP), so ``forward`` here is only the stand-alone module call.
Inference only (no backward through these modules yet).
"""
import torch

from .. import _lib


def _padded_starts(B, N, device):
    return torch.arange(0, B * N + 1, N, dtype=torch.int32, device=device)


class NearestNeighborMLP(torch.nn.Module):
    """Concatenated embeddings of the relative position (and velocity) of the n nearest neighbours
    (reference lstm/non_gridbased_pooling.py:64-147)."""

    def __init__(self, n=4, out_dim=32, no_vel=False):
        super(NearestNeighborMLP, self).__init__()
        if n < 1 or n > 8 or out_dim % n != 0:
            raise ValueError('n must divide out_dim (reference :85) and be <= 8 on the MI355X path')
        self.n = n
        self.out_dim = out_dim
        self.no_velocity = no_vel
        self.input_dim = 2 if self.no_velocity else 4
        self.embedding = torch.nn.Sequential(
            torch.nn.Linear(self.input_dim, int(out_dim / self.n)),
            torch.nn.ReLU(),)

    def reset(self, num_tracks, max_num_neigh, device):
        self.track_mask = None

    def forward(self, _, obs1, obs2):
        lin = self.embedding[0]
        dev = lin.weight.device
        _lib.require_device(lin.weight, 'NearestNeighborMLP parameters')
        B, N = obs2.size(0), obs2.size(1)
        o1 = _lib.f32c(obs1, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        out = torch.empty(B * N, self.out_dim, dtype=torch.float32, device=dev)
        starts = _padded_starts(B, N, dev)
        _lib.check(_lib.lib().tnp_pool_nn_forward(
            _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(starts), B, self.n, self.input_dim, _lib.ptr(_lib.f32c(lin.weight.detach(), dev)),
            _lib.ptr(_lib.f32c(lin.bias.detach(), dev)), self.out_dim // self.n, _lib.ptr(out), self.out_dim,
            _lib.stream_ptr()), 'tnp_pool_nn_forward')
        return out


class HiddenStateMLPPooling(torch.nn.Module):
    """Max-pooled embeddings of relative position, hidden state and relative velocity of all tracks of the scene,
    as in Social GAN (reference lstm/non_gridbased_pooling.py:150-239)."""

    def __init__(self, hidden_dim=128, mlp_dim=128, mlp_dim_spatial=32, mlp_dim_vel=32, out_dim=None):
        super(HiddenStateMLPPooling, self).__init__()
        self.out_dim = out_dim or hidden_dim
        self.hidden_dim = hidden_dim
        self.mlp_dim = mlp_dim
        self.mlp_dim_spatial = mlp_dim_spatial
        self.mlp_dim_vel = mlp_dim_vel
        self.mlp_dim_hidden = mlp_dim - mlp_dim_spatial - mlp_dim_vel
        if self.mlp_dim_hidden > 64:
            raise NotImplementedError('mlp_dim_hidden > 64 is not supported on the MI355X path')
        self.spatial_embedding = torch.nn.Sequential(
            torch.nn.Linear(2, self.mlp_dim_spatial),
            torch.nn.ReLU(),)
        if self.mlp_dim_vel:
            self.vel_embedding = torch.nn.Sequential(
                torch.nn.Linear(2, self.mlp_dim_vel),
                torch.nn.ReLU(),)
        if self.mlp_dim_hidden:
            self.hidden_embedding = torch.nn.Sequential(
                torch.nn.Linear(self.hidden_dim, self.mlp_dim_hidden),
                torch.nn.ReLU(),)
        self.out_projection = torch.nn.Linear(self.mlp_dim, self.out_dim)

    def reset(self, num_tracks, max_num_neigh, device):
        self.track_mask = None

    def forward(self, hidden_states, obs1, obs2):
        dev = self.out_projection.weight.device
        _lib.require_device(self.out_projection.weight, 'HiddenStateMLPPooling parameters')
        B, N = obs2.size(0), obs2.size(1)
        o1 = _lib.f32c(obs1, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        ms, mv, mh = self.mlp_dim_spatial, self.mlp_dim_vel, self.mlp_dim_hidden
        henc = None
        if mh:
            # embed_with_masking (reference :53-61): rows with a NaN hidden state -> fill value -100
            h = _lib.f32c(hidden_states, dev).reshape(B * N, -1)
            nan_rows = torch.isnan(h).any(dim=1, keepdim=True)
            lin = self.hidden_embedding[0]
            henc = _lib.linear_forward(torch.nan_to_num(h), lin.weight.detach(), lin.bias.detach(), relu=True)
            henc = torch.where(nan_rows, torch.full_like(henc, -100.0), henc).contiguous()
        pooled = torch.empty(B * N, ms + mh + mv, dtype=torch.float32, device=dev)
        sp = self.spatial_embedding[0]
        ve = self.vel_embedding[0] if mv else None
        starts = _padded_starts(B, N, dev)
        _lib.check(_lib.lib().tnp_pool_hiddenmlp_forward(
            _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(henc), mh, 0, _lib.ptr(starts), B, ms, mv, mh,
            _lib.ptr(_lib.f32c(sp.weight.detach(), dev)), _lib.ptr(_lib.f32c(sp.bias.detach(), dev)),
            _lib.ptr(_lib.f32c(ve.weight.detach(), dev)) if ve is not None else None,
            _lib.ptr(_lib.f32c(ve.bias.detach(), dev)) if ve is not None else None,
            _lib.ptr(pooled), ms + mh + mv, _lib.stream_ptr()), 'tnp_pool_hiddenmlp_forward')
        return _lib.linear_forward(pooled, self.out_projection.weight.detach(), self.out_projection.bias.detach())


class AttentionMLPPooling(torch.nn.Module):
    """Attention-weighted embeddings of relative position, hidden state and relative velocity of all slots of the
    padded scene, as in S-BiGAT (reference lstm/non_gridbased_pooling.py:242-351).  Same parameters / state_dict keys
    as the reference (``wq wk wv`` without bias, a single-head ``torch.nn.MultiheadAttention``, ``out_projection``);
    the module objects only hold the parameters, the computation runs in csrc/pool_nongrid.hip with the linear maps
    around the softmax folded into three small matrices (``folded()``)."""

    def __init__(self, hidden_dim=128, mlp_dim=128, mlp_dim_spatial=32, mlp_dim_vel=32, out_dim=None, fill_value=-10):
        super(AttentionMLPPooling, self).__init__()
        self.out_dim = out_dim or hidden_dim
        self.hidden_dim = hidden_dim
        self.fill_value = fill_value
        self.mlp_dim = mlp_dim
        self.mlp_dim_spatial = mlp_dim_spatial
        self.mlp_dim_vel = mlp_dim_vel
        self.mlp_dim_hidden = mlp_dim - mlp_dim_spatial - mlp_dim_vel
        if self.mlp_dim_hidden > 64 or mlp_dim > 256:
            raise NotImplementedError('mlp_dim_hidden > 64 / mlp_dim > 256 is not supported on the MI355X path')
        self.spatial_embedding = torch.nn.Sequential(
            torch.nn.Linear(2, self.mlp_dim_spatial),
            torch.nn.ReLU(),)
        if self.mlp_dim_vel:
            self.vel_embedding = torch.nn.Sequential(
                torch.nn.Linear(2, self.mlp_dim_vel),
                torch.nn.ReLU(),)
        if self.mlp_dim_hidden:
            self.hidden_embedding = torch.nn.Sequential(
                torch.nn.Linear(self.hidden_dim, self.mlp_dim_hidden),
                torch.nn.ReLU(),)
        self.wq = torch.nn.Linear(self.mlp_dim, self.mlp_dim, bias=False)
        self.wk = torch.nn.Linear(self.mlp_dim, self.mlp_dim, bias=False)
        self.wv = torch.nn.Linear(self.mlp_dim, self.mlp_dim, bias=False)
        self.multihead_attn = torch.nn.MultiheadAttention(embed_dim=self.mlp_dim, num_heads=1)
        self.out_projection = torch.nn.Linear(self.mlp_dim, self.out_dim)
        self._folded = None     # cache of the folded attention maps: rebuilt lazily, not pickled

    def reset(self, num_tracks, max_num_neigh, device):
        self.track_mask = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_folded'] = None
        return state

    def folded(self):
        """(Wq [D,D], bq [D], Wu [D+4,D], Wfin [P,D], bfin [P]) on the parameters' device, recomputed when a parameter
        changes: q = Wq e + bq;  u = Wu q with rows 0..D-1 = (in_proj_k wk)^T and row D = in_proj_bias_k;
        pooled = Wfin (sum_j a_j e_j) + bfin  (the attention weights sum to one, so the value / output biases fold)."""
        params = [self.wq.weight, self.wk.weight, self.wv.weight, self.multihead_attn.in_proj_weight,
                  self.multihead_attn.in_proj_bias, self.multihead_attn.out_proj.weight,
                  self.multihead_attn.out_proj.bias, self.out_projection.weight, self.out_projection.bias]
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        if self._folded is None or self._folded[0] != key:
            D = self.mlp_dim
            wq, wk, wv, w_in, b_in, wo, bo, wop, bop = (p.detach().double() for p in params)
            wq_eff = w_in[:D] @ wq
            wk_eff = w_in[D:2 * D] @ wk
            wv_eff = w_in[2 * D:] @ wv
            wu = torch.zeros(D + 4, D, dtype=torch.float64, device=wq.device)
            wu[:D] = wk_eff.t()
            wu[D] = b_in[D:2 * D]
            wfin = wop @ wo @ wv_eff
            bfin = wop @ (wo @ b_in[2 * D:] + bo) + bop
            self._folded = (key, tuple(t.float().contiguous() for t in (wq_eff, b_in[:D], wu, wfin, bfin)))
        return self._folded[1]

    def forward(self, hidden_states, obs1, obs2):
        dev = self.out_projection.weight.device
        _lib.require_device(self.out_projection.weight, 'AttentionMLPPooling parameters')
        B, N = obs2.size(0), obs2.size(1)
        o1 = _lib.f32c(obs1, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        ms, mv, mh = self.mlp_dim_spatial, self.mlp_dim_vel, self.mlp_dim_hidden
        D = self.mlp_dim
        henc = None
        if mh:   # embed_with_masking(..., fill_value=0) (reference :327)
            h = _lib.f32c(hidden_states, dev).reshape(B * N, -1)
            nan_rows = torch.isnan(h).any(dim=1, keepdim=True)
            lin = self.hidden_embedding[0]
            henc = _lib.linear_forward(torch.nan_to_num(h), lin.weight.detach(), lin.bias.detach(), relu=True)
            henc = torch.where(nan_rows, torch.zeros_like(henc), henc).contiguous()
        sp = self.spatial_embedding[0]
        ve = self.vel_embedding[0] if mv else None
        f = lambda t: _lib.ptr(_lib.f32c(t.detach(), dev)) if t is not None else None
        wq, bq, wu, wfin, bfin = self.folded()
        L = _lib.lib()
        e_self = torch.empty(B * N, D, dtype=torch.float32, device=dev)
        _lib.check(L.tnp_pool_attn_self(_lib.ptr(o1), _lib.ptr(o2), _lib.ptr(henc), mh, 0, B * N, ms, mv, mh, f(sp.bias),
                                        f(ve.bias) if ve is not None else None, float(self.fill_value), _lib.ptr(e_self), D,
                                        _lib.stream_ptr()), 'tnp_pool_attn_self')
        q = _lib.linear_forward(e_self, wq, bq)
        u = _lib.linear_forward(q, wu, None)
        ebar = torch.empty(B * N, D, dtype=torch.float32, device=dev)
        starts = _padded_starts(B, N, dev)
        _lib.check(L.tnp_pool_attn_pair(_lib.ptr(o1), _lib.ptr(o2), _lib.ptr(henc), mh, 0, _lib.ptr(starts), B, N, None, ms, mv, mh,
                                        f(sp.weight), f(sp.bias), f(ve.weight) if ve is not None else None,
                                        f(ve.bias) if ve is not None else None, float(self.fill_value), _lib.ptr(u), D + 4,
                                        _lib.ptr(ebar), D, _lib.stream_ptr()), 'tnp_pool_attn_pair')
        return _lib.linear_forward(ebar, wfin, bfin)


class _StatefulInteractionEncoder(torch.nn.Module):
    """Shared part of NearestNeighborLSTM / TrajectronPooling: the interaction-encoder ``pool_lstm`` whose state lives
    per padded slot and is zeroed by ``reset`` at the start of every ``LSTM.forward`` (reference :385-389, :483-487).
    Inside ``LSTM.forward`` the state is kept by the fused sequence driver; the stand-alone ``forward`` below keeps it
    in ``self.hidden_cell_state`` as the reference does."""

    def _init_encoder(self, hidden_dim, out_dim):
        self.hidden_dim = hidden_dim
        self.pool_lstm = torch.nn.LSTMCell(out_dim, hidden_dim)
        self.hidden2pool = torch.nn.Linear(hidden_dim, out_dim)
        self.hidden_cell_state = None

    def reset(self, num_tracks, max_num_neigh, device):
        dev = self.hidden2pool.weight.device
        self.hidden_cell_state = [torch.zeros(num_tracks, self.hidden_dim, device=dev),
                                  torch.zeros(num_tracks, self.hidden_dim, device=dev)]

    def _encode(self, feat):
        """pool_lstm + hidden2pool on the feature rows (GEMMs on the matrix cores, gate nonlinearities pointwise)."""
        if self.hidden_cell_state is None or self.hidden_cell_state[0].size(0) != feat.size(0):
            self.reset(feat.size(0), 0, feat.device)
        h, c = self.hidden_cell_state
        cell = self.pool_lstm
        gates = _lib.linear_forward(feat, cell.weight_ih.detach(), cell.bias_ih.detach()) + \
            _lib.linear_forward(h, cell.weight_hh.detach(), cell.bias_hh.detach())
        H = self.hidden_dim
        i, f, g, o = torch.sigmoid(gates[:, :H]), torch.sigmoid(gates[:, H:2 * H]), torch.tanh(gates[:, 2 * H:3 * H]), \
            torch.sigmoid(gates[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        self.hidden_cell_state = [h, c]
        return _lib.linear_forward(h, self.hidden2pool.weight.detach(), self.hidden2pool.bias.detach())


class NearestNeighborLSTM(_StatefulInteractionEncoder):
    """NearestNeighborMLP features passed through an interaction-encoder LSTM
    (reference lstm/non_gridbased_pooling.py:354-455)."""

    def __init__(self, n=4, hidden_dim=256, out_dim=32):
        super(NearestNeighborLSTM, self).__init__()
        if n < 1 or n > 8 or out_dim % n != 0 or out_dim % 4 != 0 or hidden_dim % 32 != 0:
            raise ValueError('n must divide out_dim (n <= 8, out_dim % 4 == 0, hidden_dim % 32 == 0 on the MI355X path)')
        self.n = n
        self.out_dim = out_dim
        self.input_dim = 4
        self.embedding = torch.nn.Sequential(
            torch.nn.Linear(self.input_dim, int(out_dim / self.n)),
            torch.nn.ReLU(),)
        self._init_encoder(hidden_dim, out_dim)

    def forward(self, _, obs1, obs2):
        lin = self.embedding[0]
        dev = lin.weight.device
        _lib.require_device(lin.weight, 'NearestNeighborLSTM parameters')
        B, N = obs2.size(0), obs2.size(1)
        o1 = _lib.f32c(obs1, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        feat = torch.empty(B * N, self.out_dim, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().tnp_pool_nn_forward(
            _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(_padded_starts(B, N, dev)), B, self.n, 4,
            _lib.ptr(_lib.f32c(lin.weight.detach(), dev)), _lib.ptr(_lib.f32c(lin.bias.detach(), dev)),
            self.out_dim // self.n, _lib.ptr(feat), self.out_dim, _lib.stream_ptr()), 'tnp_pool_nn_forward')
        return self._encode(feat)


class TrajectronPooling(_StatefulInteractionEncoder):
    """Own state ++ sum of the other visible tracks' states, embedded and passed through an interaction-encoder LSTM
    (reference lstm/non_gridbased_pooling.py:457-538).  As in the reference the sum runs over all visible tracks of
    the BATCH, not of the scene."""

    def __init__(self, n=4, hidden_dim=256, out_dim=32, track_mask=None):
        super(TrajectronPooling, self).__init__()
        if out_dim % 4 != 0 or hidden_dim % 32 != 0:
            raise ValueError('out_dim % 4 == 0 and hidden_dim % 32 == 0 required on the MI355X path')
        self.n = n
        self.out_dim = out_dim
        self.embedding = torch.nn.Sequential(
            torch.nn.Linear(8, out_dim),
            torch.nn.ReLU(),)
        self._init_encoder(hidden_dim, out_dim)
        self.track_mask = track_mask

    def forward(self, _, obs1, obs2):
        lin = self.embedding[0]
        dev = lin.weight.device
        _lib.require_device(lin.weight, 'TrajectronPooling parameters')
        B, N = obs2.size(0), obs2.size(1)
        o1 = _lib.f32c(obs1, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        feat = torch.empty(B * N, self.out_dim, dtype=torch.float32, device=dev)
        scratch = torch.empty(4, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().tnp_pool_traj_forward(
            _lib.ptr(o1), _lib.ptr(o2), B * N, _lib.ptr(_lib.f32c(lin.weight.detach(), dev)),
            _lib.ptr(_lib.f32c(lin.bias.detach(), dev)), self.out_dim, _lib.ptr(feat), self.out_dim, _lib.ptr(scratch),
            _lib.stream_ptr()), 'tnp_pool_traj_forward')
        return self._encode(feat)
