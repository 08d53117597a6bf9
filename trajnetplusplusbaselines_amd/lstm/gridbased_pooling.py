"""GridBasedPooling on MI355X: same constructor, attributes, public methods and state_dict keys as the
reference's lstm/gridbased_pooling.py:15-400; the grid build and the embedding MLP run in hand-written HIP
(csrc/pool_grid.hip, csrc/gemm_f32_mfma.hip).  No CPU fallback."""
import ctypes

import torch

from .. import _lib


class _GridScatter(torch.autograd.Function):
    """Autograd through the STAND-ALONE grid build (reference ``pool(hidden_state, obs1, obs2)`` is differentiable,
    lstm/gridbased_pooling.py:94-110, 227-305): ``occ[arange, oi] = other_values`` hands every in-range neighbour the gradient of
    its cell (overwritten duplicates included), and the grid leaves ``occupancy()`` through ``lp_pool2d(x, 1, 1)`` whose
    derivative at 0 is 0 (:304).  The tables are ``tnp_pool_pair_cells_autograd``'s -- the ones the training path of
    ``LSTM.forward`` uses -- and the reductions ``tnp_social_scatter_backward`` / ``tnp_directional_scatter_backward``.
    Cell indices are not differentiable (the reference's ``.long()``, :273)."""

    @staticmethod
    def forward(ctx, pool, type_id, obs1, obs2, values):
        grid, _ = pool._winner_grid(obs1, obs2, type_id, values=values)
        ctx.pool, ctx.type_id, ctx.shape = pool, type_id, (obs2.size(0), obs2.size(1))
        ctx.save_for_backward(obs1, obs2)
        return grid

    @staticmethod
    def backward(ctx, d_grid):
        pool, (B, N) = ctx.pool, ctx.shape
        obs1, obs2 = ctx.saved_tensors
        dev = d_grid.device
        G, cell, half_x, half_y = pool._geometry()
        M, ncell = B * N, G * G
        L = _lib.lib()
        o1 = _lib.f32c(obs1 if obs1 is not None else obs2, dev).reshape(M, 2)
        o2 = _lib.f32c(obs2, dev).reshape(M, 2)
        row_base = (torch.arange(M, dtype=torch.int32, device=dev) // N) * N
        row_count = torch.full((M,), N, dtype=torch.int32, device=dev)
        raw = torch.empty(M, N, dtype=torch.int32, device=dev)
        cells = torch.empty(M, N, dtype=torch.int32, device=dev)
        directional = ctx.type_id == _lib.POOL_DIRECTIONAL
        winner = torch.empty(M, N, dtype=torch.int32, device=dev) if directional else None
        _lib.check(L.tnp_pool_pair_cells_autograd(_lib.ptr(o2), _lib.ptr(row_base), _lib.ptr(row_count), None, N, M, N, G, cell,
                                                  half_x, half_y, float(pool.constant), _lib.ptr(raw), _lib.ptr(cells),
                                                  _lib.ptr(winner), _lib.stream_ptr()), 'tnp_pool_pair_cells_autograd')
        dg = _lib.f32c(d_grid, dev)
        d_obs1 = d_obs2 = d_values = None
        if directional:
            dvel = torch.empty(M, 2, dtype=torch.float32, device=dev)
            _lib.check(L.tnp_directional_scatter_backward(_lib.ptr(dg), dg.stride(0), _lib.ptr(cells), _lib.ptr(winner),
                                                          _lib.ptr(row_base), _lib.ptr(row_count), _lib.ptr(o1), _lib.ptr(o2), M, N,
                                                          ncell, _lib.ptr(dvel), _lib.stream_ptr()), 'tnp_directional_scatter_backward')
            dvel = dvel.reshape(B, N, 2)
            if ctx.needs_input_grad[3]:
                d_obs2 = dvel
            if ctx.needs_input_grad[2]:
                d_obs1 = -dvel
        elif ctx.needs_input_grad[4]:
            C = dg.size(1) // ncell
            d_values = torch.empty(M, C, dtype=torch.float32, device=dev)
            _lib.check(L.tnp_social_scatter_backward(_lib.ptr(dg), dg.stride(0), _lib.ptr(cells), _lib.ptr(row_base),
                                                     _lib.ptr(row_count), M, N, C, ncell, _lib.ptr(d_values), _lib.stream_ptr()),
                       'tnp_social_scatter_backward')
        return None, None, d_obs1, d_obs2, d_values


class GridBasedPooling(torch.nn.Module):
    def __init__(self, cell_side=2.0, n=4, hidden_dim=128, out_dim=None,
                 type_='occupancy', pool_size=1, blur_size=1, front=False,
                 embedding_arch='one_layer', pretrained_pool_encoder=None,
                 constant=0, norm=0, layer_dims=None, latent_dim=16):
        """Pools in a grid of size 'n * cell_side' centred at the ped location
        (arguments as in reference lstm/gridbased_pooling.py:16-41)."""
        super(GridBasedPooling, self).__init__()
        self.cell_side = cell_side
        self.n = n
        self.type_ = type_
        self.pool_size = pool_size
        self.blur_size = blur_size
        self.norm_pool = False
        self.front = front
        if self.front:
            self.norm_pool = True
        self.constant = constant
        self.norm = norm
        self.pool_scale = 1.0

        if self.type_ not in ('occupancy', 'directional', 'social'):
            # reference 'dir_social' concatenates along the neighbour axis (gridbased_pooling.py:209) and only
            # runs for latent_dim == 2; it is not part of any BASELINE config
            raise NotImplementedError("GridBasedPooling type_ %r is not supported on the MI355X path" % (type_,))
        self.pooling_dim = 1
        if self.type_ == 'directional':
            self.pooling_dim = 2
        if self.type_ == 'social':
            self.hidden_dim_encoding = torch.nn.Linear(hidden_dim, latent_dim)
            self.pooling_dim = latent_dim

        if out_dim is None:
            out_dim = hidden_dim
        self.out_dim = out_dim

        if pretrained_pool_encoder is not None:
            raise NotImplementedError('pretrained_pool_encoder is not supported on the MI355X path')
        self.pretrained_model = None

        self.embedding = None
        self.embedding_arch = embedding_arch
        if self.embedding_arch == 'one_layer':
            self.embedding = self.one_layer()
        elif self.embedding_arch == 'two_layer':
            self.embedding = self.two_layer(None, layer_dims)
        elif self.embedding_arch == 'three_layer':
            self.embedding = self.three_layer(None, layer_dims)
        elif self.embedding_arch == 'lstm_layer':
            self.embedding = self.lstm_layer(hidden_dim)

    # ---- embedding architectures (reference :308-335) --------------------------------------------
    def one_layer(self, input_dim=None):
        if input_dim is None:
            input_dim = self.n * self.n * self.pooling_dim
        return torch.nn.Sequential(
            torch.nn.Linear(input_dim, self.out_dim),
            torch.nn.ReLU(),)

    def two_layer(self, input_dim=None, layer_dims=None):
        if input_dim is None:
            input_dim = self.n * self.n * self.pooling_dim
        return torch.nn.Sequential(
            torch.nn.Linear(input_dim, layer_dims[0]),
            torch.nn.ReLU(),
            torch.nn.Linear(layer_dims[0], self.out_dim),
            torch.nn.ReLU(),)

    def three_layer(self, input_dim=None, layer_dims=None):
        if input_dim is None:
            input_dim = self.n * self.n * self.pooling_dim
        return torch.nn.Sequential(
            torch.nn.Linear(input_dim, layer_dims[0]),
            torch.nn.ReLU(),
            torch.nn.Linear(layer_dims[0], layer_dims[1]),
            torch.nn.ReLU(),
            torch.nn.Linear(layer_dims[1], self.out_dim),
            torch.nn.ReLU(),)

    def lstm_layer(self, hidden_dim):
        """reference :337-343.  `forward` only ever applies the returned Linear + ReLU (:107-109): `lstm_forward`
        (:353-379) is never called, so `pool_lstm` / `hidden2pool` exist for state_dict compatibility (and are created
        first, so that seeded initialisation matches the reference) and the arch behaves like 'one_layer'."""
        self.hidden_dim = hidden_dim
        self.pool_lstm = torch.nn.LSTMCell(self.out_dim, self.hidden_dim)
        self.hidden2pool = torch.nn.Linear(self.hidden_dim, self.out_dim)
        return torch.nn.Sequential(
            torch.nn.Linear(self.n * self.n * self.pooling_dim, self.out_dim),
            torch.nn.ReLU(),)

    def reset(self, num_tracks, max_num_neigh, device):
        """Called once per forward by the reference (lstm/lstm.py:213-216); nothing to reset here (the reference's
        'lstm_layer' state, :347-351, is never read by its forward)."""
        self.track_mask = None

    # ---- helpers -----------------------------------------------------------------------------------
    def embedding_layers(self):
        if self.embedding is None:
            return []
        return [m for m in self.embedding if isinstance(m, torch.nn.Linear)]

    def _geometry(self):
        G = self.n * self.pool_size
        cell = float(torch.tensor(self.cell_side / self.pool_size, dtype=torch.float32))
        half = float(G / 2)
        return G, cell, half, (0.0 if self.front else half)

    def _device(self, obs):
        for p in self.parameters():
            return p.device
        return obs.device

    def _winner_grid(self, obs1, obs2, type_id, values=None, want_grid=True, want_winners=False):
        """obs1/obs2 [B,N,2] padded tensors -> dense grid [B*N, C*G*G] and/or winner table [B*N, G*G]."""
        dev = obs2.device
        _lib.require_device(obs2, 'obs2')
        B, N = obs2.size(0), obs2.size(1)
        G, cell, half_x, half_y = self._geometry()
        C = {_lib.POOL_OCCUPANCY: 1, _lib.POOL_DIRECTIONAL: 2}.get(type_id, self.pooling_dim)
        o1 = _lib.f32c(obs1 if obs1 is not None else obs2, dev).reshape(B * N, 2)
        o2 = _lib.f32c(obs2, dev).reshape(B * N, 2)
        starts = torch.arange(0, B * N + 1, N, dtype=torch.int32, device=dev)
        grid = torch.empty(B * N, C * G * G, dtype=torch.float32, device=dev) if want_grid else None
        winners = torch.empty(B * N, G * G, dtype=torch.int16, device=dev) if want_winners else None
        vals = _lib.f32c(values, dev).reshape(B * N, -1) if values is not None else None
        _lib.check(_lib.lib().tnp_pool_grid_forward(
            type_id, _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(vals), vals.stride(0) if vals is not None else 0,
            _lib.ptr(starts), B, N, None, G, C, cell, half_x, half_y, float(self.constant),
            _lib.ptr(grid), C * G * G, _lib.ptr(winners), _lib.stream_ptr()), 'tnp_pool_grid_forward')
        return grid, winners

    def _finish(self, grid, rows, C):
        """blur + pool_size reduction of the fine grid (reference :297-303).  The trainer never changes the
        defaults pool_size = blur_size = 1, for which this is the identity."""
        G = self.n * self.pool_size
        occ = grid.view(rows, C, G, G)
        if self.blur_size != 1:
            occ = torch.nn.functional.avg_pool2d(occ, self.blur_size, 1, int(self.blur_size / 2),
                                                 count_include_pad=True)
        if self.pool_size != 1:
            occ = torch.nn.functional.lp_pool2d(occ, 1, self.pool_size)
        return occ

    # ---- public grid builders (reference :112-170, 227-305) -----------------------------------------
    def occupancies(self, obs1, obs2):
        return self.occupancy(obs2, past_obs=obs1)

    def occupancy(self, obs, other_values=None, past_obs=None):
        """Occupancy map filled with `other_values` ([B,N,N-1,C], None = ones). Returns [B*N, C, n, n]."""
        B, N = obs.size(0), obs.size(1)
        if N == 1:  # reference :252-253
            return self.constant * torch.ones(1, self.pooling_dim, self.n, self.n, device=obs.device)
        if other_values is None:
            grid, _ = self._winner_grid(past_obs, obs, _lib.POOL_OCCUPANCY)
            return self._finish(grid, B * N, 1)
        # arbitrary per-pair values: build the winner table on the GPU and gather the pair values
        _, winners = self._winner_grid(past_obs, obs, _lib.POOL_OCCUPANCY, want_grid=False, want_winners=True)
        C = other_values.size(-1)
        vals = _lib.f32c(other_values, obs.device).reshape(B * N, N - 1, C)
        w = winners.long()
        ego = (torch.arange(B * N, device=obs.device) % N).unsqueeze(1)
        jj = (w - (w > ego).long()).clamp(min=0)                                      # neighbour slot j -> j'
        picked = torch.gather(vals, 1, jj.unsqueeze(-1).expand(-1, -1, C))            # [rows, G*G, C]
        bg = torch.full_like(picked, float(self.constant))
        grid = torch.where((w >= 0).unsqueeze(-1), picked, bg).transpose(1, 2).contiguous()
        return self._finish(grid.view(B * N, -1), B * N, C)

    def directional(self, obs1, obs2):
        B, N = obs2.size(0), obs2.size(1)
        if N == 1:
            return self.occupancy(obs2, None, past_obs=obs1)
        if torch.is_grad_enabled() and (obs2.requires_grad or (obs1 is not None and obs1.requires_grad)):
            grid = _GridScatter.apply(self, _lib.POOL_DIRECTIONAL, obs1, obs2, None)     # velocities carry gradient
        else:
            grid, _ = self._winner_grid(obs1, obs2, _lib.POOL_DIRECTIONAL)
        return self._finish(grid, B * N, 2)

    def social(self, hidden_state, obs1, obs2):
        B, N = obs2.size(0), obs2.size(1)
        if N == 1:
            return self.occupancy(obs2, None, past_obs=obs1)
        lin = self.hidden_dim_encoding
        _lib.require_device(lin.weight, 'GridBasedPooling parameters')
        h = torch.nan_to_num(_lib.f32c(hidden_state, lin.weight.device).reshape(B * N, -1))  # reference :166
        if torch.is_grad_enabled() and (h.requires_grad or lin.weight.requires_grad):
            # stand-alone call under autograd (outside LSTM.forward, whose sequence op has its own backward sweep): the same
            # kernels, wrapped so that gradients reach hidden_state and hidden_dim_encoding (reference :165-170)
            from .. import ops  # noqa: F401  (registers trajnet::linear with its autograd rule)
            enc = torch.ops.trajnet.linear(h, lin.weight, lin.bias, False)
            grid = _GridScatter.apply(self, _lib.POOL_SOCIAL, obs1, obs2, enc)
        else:
            enc = _lib.linear_forward(h, lin.weight.detach(), lin.bias.detach())
            grid, _ = self._winner_grid(obs1, obs2, _lib.POOL_SOCIAL, values=enc)
        return self._finish(grid, B * N, self.pooling_dim)

    def forward(self, hidden_state, obs1, obs2):
        """[B,N,H], [B,N,2], [B,N,2] -> [B*N, out_dim]  (reference :94-110)."""
        batch_size, num_tracks = obs1.size(0), obs1.size(1)
        if self.type_ == 'occupancy':
            grid = self.occupancies(obs1, obs2)
        elif self.type_ == 'directional':
            grid = self.directional(obs1, obs2)
        else:
            grid = self.social(hidden_state, obs1, obs2)
        grid = grid.reshape(batch_size * num_tracks, -1)
        x = grid
        layers = self.embedding_layers()
        if torch.is_grad_enabled() and (x.requires_grad or any(l.weight.requires_grad for l in layers)):
            from .. import ops  # noqa: F401
            for lin in layers:     # trajnet::linear = the same MFMA GEMM + its data / weight gradient kernels
                x = torch.ops.trajnet.linear(x, lin.weight, lin.bias, True)
            return x
        for lin in layers:
            x = _lib.linear_forward(x, lin.weight.detach(), lin.bias.detach(), relu=True)
        return x

    def make_grid(self, obs):
        """Grids for all time-steps (occupancy / directional only), reference :381-400."""
        if obs.ndim == 2:
            obs = obs.unsqueeze(0)
        grid = []
        for i in range(1, obs.size(0)):
            obs1, obs2 = obs[i - 1], obs[i]
            track_mask = (torch.isnan(obs1[:, 0]) + torch.isnan(obs2[:, 0])) == 0
            obs1, obs2 = obs1[track_mask], obs2[track_mask]
            if self.type_ == 'occupancy':
                grid.append(self.occupancies(obs1.unsqueeze(0), obs2.unsqueeze(0)))
            elif self.type_ == 'directional':
                grid.append(self.directional(obs1.unsqueeze(0), obs2.unsqueeze(0)))
        return grid
