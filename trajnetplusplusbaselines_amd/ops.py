"""``torch.library`` registration of the C-ABI entry points ("each registered as a torch.library op taking /
returning torch.Tensor on the current HIP stream, no host sync, errors surfaced as exceptions").

The ops live in the ``trajnet`` namespace (``torch.ops.trajnet.linear`` ...).  Each has a fake (meta) implementation, so
``torch.compile`` / ``FakeTensorMode`` can trace through them, and ``trajnet::linear`` carries its autograd formula.  The
recurrent sequence itself is ``trajnet::lstm_sequence`` for inference and ``trajnet::lstm_sequence_train`` +
``trajnet::lstm_sequence_backward`` for training (``LSTM.forward`` routes through them while ``torch.compile`` traces, so a
model compiles without a graph break in either mode).  The training pair wraps the same forward / backward code as the
eager ``torch.autograd.Function`` (lstm/training.py); what that Function keeps on its ctx -- per-step buffers, index
objects, not tensors an op schema can describe -- stays in this module behind an integer handle that travels through the
graph as a tensor.  The implementations are the same ctypes calls the
module classes make -- the ops add dispatcher visibility, not another code path -- and there is no CPU kernel: a host
tensor raises, as everywhere in this package.

    pool_grid_winners(obs1, obs2, scene_start, n_max, n, cell_side)                     -> int16 [M, n*n]
    pool_grid(type, obs1, obs2, values?, scene_start, n_max, n, cell_side, constant)    -> fp32 [M, C*n*n]
    linear(x, weight, bias?, relu)                                                      -> fp32 [M, N]        (+ autograd)
    pool_embed_sparse(winners, values, scene_start, weight, bias?, relu)                -> fp32 [M, N1]
    constant_velocity(last, prev, n_predict)                                            -> fp64 [n_predict, N, 2]
    sf_rollout(state, scene_start, n_max, n_predict, v0, sigma, tau)                    -> fp64 [n_predict, M, 2]
    lstm_sequence(observed, goals?, batch_split, truth?, t_dec, pad_to, model, params)  -> (fp32 [S, M, 5], fp32 [S', M, 2])
    lstm_sequence_train(same arguments)             -> (fp32 [S, M, 5], fp32 [S', M, 2], fp32 [M, H], int64 handle)   (+ autograd)
    lstm_sequence_backward(handle, d_rel, d_pred, d_hlast, params)                      -> list of parameter gradients
"""
import collections
import itertools
import weakref
from typing import List, Optional, Tuple

import torch

from . import _lib

_POOL = {'occupancy': _lib.POOL_OCCUPANCY, 'directional': _lib.POOL_DIRECTIONAL, 'social': _lib.POOL_SOCIAL}


def _dev(t, what):
    _lib.require_device(t, what)
    return t.device


def _starts(scene_start, dev):
    return scene_start.to(device=dev, dtype=torch.int32).contiguous()


@torch.library.custom_op('trajnet::pool_grid_winners', mutates_args=())
def pool_grid_winners(obs1: torch.Tensor, obs2: torch.Tensor, scene_start: torch.Tensor, n_max: int, n: int,
                      cell_side: float) -> torch.Tensor:
    """Winner table of the occupancy map (reference lstm/gridbased_pooling.py:227-305): for every track and cell the
    scene-local index of the neighbour whose value the grid holds (last writer wins), -1 = empty."""
    dev = _dev(obs2, 'obs2')
    o1, o2 = _lib.f32c(obs1, dev).reshape(-1, 2), _lib.f32c(obs2, dev).reshape(-1, 2)
    st = _starts(scene_start, dev)
    M = o2.shape[0]
    win = torch.empty(M, n * n, dtype=torch.int16, device=dev)
    _lib.check(_lib.lib().tnp_pool_grid_forward(_lib.POOL_OCCUPANCY, _lib.ptr(o1), _lib.ptr(o2), None, 0, _lib.ptr(st),
                                                st.numel() - 1, int(n_max), None, int(n), 1, float(cell_side), n / 2, n / 2,
                                                0.0, None, 0, _lib.ptr(win), _lib.stream_ptr()), 'tnp_pool_grid_forward')
    return win


@pool_grid_winners.register_fake
def _(obs1, obs2, scene_start, n_max, n, cell_side):
    return obs2.new_empty((obs2.reshape(-1, 2).shape[0], n * n), dtype=torch.int16)


@torch.library.custom_op('trajnet::pool_grid', mutates_args=())
def pool_grid(type_: str, obs1: torch.Tensor, obs2: torch.Tensor, values: Optional[torch.Tensor],
              scene_start: torch.Tensor, n_max: int, n: int, cell_side: float, constant: float) -> torch.Tensor:
    """Dense grid of GridBasedPooling.occupancies / directional / social (lstm/gridbased_pooling.py:112-170):
    [M, C*n*n] with C = 1 / 2 / values.shape[1]."""
    dev = _dev(obs2, 'obs2')
    o1, o2 = _lib.f32c(obs1, dev).reshape(-1, 2), _lib.f32c(obs2, dev).reshape(-1, 2)
    st = _starts(scene_start, dev)
    M = o2.shape[0]
    vals = _lib.f32c(values, dev).reshape(M, -1) if values is not None else None
    C = {'occupancy': 1, 'directional': 2}.get(type_, vals.shape[1] if vals is not None else 0)
    if type_ not in _POOL or C <= 0:
        raise ValueError('pool_grid: type %r (social needs `values`)' % (type_,))
    grid = torch.empty(M, C * n * n, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().tnp_pool_grid_forward(_POOL[type_], _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(vals),
                                                vals.stride(0) if vals is not None else 0, _lib.ptr(st), st.numel() - 1,
                                                int(n_max), None, int(n), C, float(cell_side), n / 2, n / 2, float(constant),
                                                _lib.ptr(grid), C * n * n, None, _lib.stream_ptr()), 'tnp_pool_grid_forward')
    return grid


@pool_grid.register_fake
def _(type_, obs1, obs2, values, scene_start, n_max, n, cell_side, constant):
    M = obs2.reshape(-1, 2).shape[0]
    C = {'occupancy': 1, 'directional': 2}.get(type_, values.reshape(M, -1).shape[1] if values is not None else 1)
    return obs2.new_empty((M, C * n * n), dtype=torch.float32)


@torch.library.custom_op('trajnet::linear', mutates_args=())
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """act(x @ weight.T + bias) on the fp32 matrix cores (tnp_linear_forward): the embedding MLPs of
    lstm/gridbased_pooling.py:308-335."""
    _dev(x, 'x')
    return _lib.linear_forward(x, weight, bias, relu=relu)


@linear.register_fake
def _(x, weight, bias, relu):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, relu = inputs
    ctx.relu, ctx.has_bias = relu, bias is not None
    ctx.save_for_backward(x, weight, output)


def _linear_backward(ctx, grad):
    x, weight, out = ctx.saved_tensors
    g = grad.contiguous()
    if ctx.relu:
        g = torch.where(out > 0, g, torch.zeros_like(g))
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = _lib.linear_forward(g, weight.t().contiguous(), None)            # data gradient: the same GEMM kernel
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        L = _lib.lib()
        K, Mo, No = g.shape[0], g.shape[1], x.shape[1]
        nbytes = L.tnp_wgrad_workspace_bytes(Mo, No, K)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.device)
        dw = torch.empty(Mo, No, device=g.device)
        db = torch.empty(Mo, device=g.device) if ctx.has_bias else None
        xc = _lib.f32c(x)
        _lib.check(L.tnp_wgrad(_lib.ptr(g), g.stride(0), _lib.ptr(xc), xc.stride(0), K, Mo, No, _lib.ptr(dw), No, _lib.ptr(db),
                               _lib.ptr(ws), nbytes, _lib.stream_ptr()), 'tnp_wgrad')
    return dx, dw, db, None


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


@torch.library.custom_op('trajnet::pool_embed_sparse', mutates_args=())
def pool_embed_sparse(winners: torch.Tensor, values: torch.Tensor, scene_start: torch.Tensor, weight: torch.Tensor,
                      bias: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """First Linear of the grid embedding applied to the social grid without materialising it (tnp_pool_embed_sparse_forward):
    ``weight`` is the module's parameter [N1, C*n*n]; the cell-major copy the kernel streams is made here."""
    dev = _dev(values, 'values')
    M, C = values.shape
    ncell, N1 = winners.shape[1], weight.shape[0]
    L = _lib.lib()
    st = _starts(scene_start, dev)
    row_base = torch.empty(M, dtype=torch.int32, device=dev)
    _lib.check(L.tnp_row_base(_lib.ptr(st), st.numel() - 1, _lib.ptr(row_base), _lib.stream_ptr()), 'tnp_row_base')
    wcm = _lib.f32c(weight, dev).view(N1, C, ncell).permute(2, 1, 0).contiguous()
    vals, bt = _lib.f32c(values, dev), (_lib.f32c(bias, dev) if bias is not None else None)
    win = winners.to(device=dev, dtype=torch.int16).contiguous()
    need = L.tnp_pool_embed_sparse_workspace_bytes(M, N1, ncell)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    out = torch.empty(M, N1, dtype=torch.float32, device=dev)
    _lib.check(L.tnp_pool_embed_sparse_forward(_lib.ptr(win), _lib.ptr(vals), vals.stride(0), _lib.ptr(row_base), _lib.ptr(wcm),
                                               _lib.ptr(bt), M, ncell, C, N1, int(relu), _lib.ptr(out), N1, _lib.ptr(ws), need,
                                               _lib.stream_ptr()), 'tnp_pool_embed_sparse_forward')
    return out


@pool_embed_sparse.register_fake
def _(winners, values, scene_start, weight, bias, relu):
    return values.new_empty((values.shape[0], weight.shape[0]), dtype=torch.float32)


@torch.library.custom_op('trajnet::constant_velocity', mutates_args=())
def constant_velocity(last: torch.Tensor, prev: torch.Tensor, n_predict: int) -> torch.Tensor:
    """classical/constant_velocity.py:4-20 on fp64 positions [N, 2] -> [n_predict, N, 2]."""
    dev = _dev(last, 'last')
    a, b = last.to(torch.float64).contiguous(), prev.to(device=dev, dtype=torch.float64).contiguous()
    N = a.shape[0]
    out = torch.empty(n_predict, N, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_constant_velocity(_lib.ptr(a), _lib.ptr(b), N, int(n_predict), _lib.ptr(out), _lib.stream_ptr()),
               'tnp_constant_velocity')
    return out


@constant_velocity.register_fake
def _(last, prev, n_predict):
    return last.new_empty((n_predict, last.shape[0], 2), dtype=torch.float64)


@torch.library.custom_op('trajnet::sf_rollout', mutates_args=())
def sf_rollout(state: torch.Tensor, scene_start: torch.Tensor, n_max: int, n_predict: int, v0: float, sigma: float,
               tau: float) -> torch.Tensor:
    """Social-force rollout of many scenes (classical/socialforce.py:84-95): state [M, 6] fp64 (x, y, vx, vy, goal),
    one output row every 8 simulator steps of 1/20 s."""
    dev = _dev(state, 'state')
    s0 = state.to(torch.float64).contiguous()
    st = _starts(scene_start, dev)
    M = s0.shape[0]
    out = torch.empty(n_predict, M, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_sf_rollout(_lib.ptr(s0), _lib.ptr(st), st.numel() - 1, M, int(n_max), int(n_predict) * 8, 8, float(v0),
                                         float(sigma), float(tau), 1.0 / 20, _lib.ptr(out), _lib.stream_ptr()), 'tnp_sf_rollout')
    return out


@sf_rollout.register_fake
def _(state, scene_start, n_max, n_predict, v0, sigma, tau):
    return state.new_empty((n_predict, state.shape[0], 2), dtype=torch.float64)


@torch.library.custom_op('trajnet::orca_rollout', mutates_args=())
def orca_rollout(pos: torch.Tensor, vel: torch.Tensor, goals: torch.Tensor, speed: torch.Tensor, scene_start: torch.Tensor,
                 n_max: int, n_predict: int, neighbor_dist: float, time_horizon: float, radius: float) -> torch.Tensor:
    """ORCA rollout of many scenes (classical/orca.py:90-119: RVO2 with 10 neighbours, time step 1/20 s, max speed 1.3 x the
    initial speed, preferred velocity towards the goal capped at the initial speed; 8 n_predict + 1 simulator steps, one output
    row every 8): pos, vel [M, 2], goals [M, 2], speed [M] -> [n_predict, M, 2] float32."""
    dev = _dev(pos, 'pos')
    p0, v0 = _lib.f32c(pos, dev), _lib.f32c(vel, dev)
    g, sp = goals.to(device=dev, dtype=torch.float64).contiguous(), speed.to(device=dev, dtype=torch.float64).contiguous()
    vmax = (1.3 * sp).to(torch.float32)
    st = _starts(scene_start, dev)
    M = p0.shape[0]
    out = torch.empty(n_predict, M, 2, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().tnp_orca_rollout(_lib.ptr(p0), _lib.ptr(v0), _lib.ptr(g), _lib.ptr(sp), _lib.ptr(vmax), _lib.ptr(st),
                                           st.numel() - 1, M, int(n_max), 8 * int(n_predict) + 1, 8, 1.0 / 20, float(neighbor_dist), 10,
                                           float(time_horizon), float(radius), _lib.ptr(out), None, _lib.stream_ptr()),
               'tnp_orca_rollout')
    return out


@orca_rollout.register_fake
def _(pos, vel, goals, speed, scene_start, n_max, n_predict, neighbor_dist, time_horizon, radius):
    return pos.new_empty((n_predict, pos.shape[0], 2), dtype=torch.float32)


@torch.library.custom_op('trajnet::kalman_predict', mutates_args=())
def kalman_predict(obs: torch.Tensor, noise: torch.Tensor, n_iter: int) -> torch.Tensor:
    """Constant-velocity Kalman predictor (classical/kalman.py:22-60): EM (``n_iter`` iterations; the reference: 10) on the
    observed track, then the mean of ``noise.shape[1]`` sampled continuations from the last smoothed state.  obs [N, T, 2]
    float64 complete tracks, noise [N, n_samples, n_steps, 6] standard-normal draws (4 state + 2 observation components per
    step) -> [N, n_steps, 2] float64 (row 0 = the last observed step, as pykalman's sample(initial_state=) returns it)."""
    dev = _dev(obs, 'obs')
    o = obs.to(torch.float64).contiguous()
    z = noise.to(device=dev, dtype=torch.float64).contiguous()
    N, T = o.shape[0], o.shape[1]
    n_samples, n_steps = z.shape[1], z.shape[2]
    out = torch.empty(N, n_steps, 2, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().tnp_kalman_predict(_lib.ptr(o), N, T, int(n_iter), n_steps, n_samples, _lib.ptr(z), 1e-5, 0.05 ** 2,
                                             _lib.ptr(out), _lib.stream_ptr()), 'tnp_kalman_predict')
    return out


@kalman_predict.register_fake
def _(obs, noise, n_iter):
    return obs.new_empty((obs.shape[0], noise.shape[2], 2), dtype=torch.float64)


# ---- the recurrent sequence (inference) ----------------------------------------------------------------------------------
# An op schema carries tensors and scalars; the model's hyper-parameters (interaction module, grid size, ...) and cached
# device buffers live in the module.  The op therefore takes an integer handle of the module (a weak registry: the handle
# dies with the model) next to the module's parameters as a tensor list, which is what makes the data dependence visible.
_MODELS = weakref.WeakValueDictionary()


def model_handle(model) -> int:
    h = id(model)
    _MODELS[h] = model
    return h


@torch.library.custom_op('trajnet::lstm_step', mutates_args=())
def lstm_step(h: torch.Tensor, c: torch.Tensor, obs1: torch.Tensor, obs2: torch.Tensor, goals: Optional[torch.Tensor],
              batch_split: torch.Tensor, decoder: bool, pad_to: int, model: int,
              params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One masked recurrent step, LSTM.step (reference lstm/lstm.py:91-168), without gradients: dense state h, c [M, H], the two
    last positions obs1, obs2 [M, 2] (NaN = absent: the track's state passes through) -> (h', c', normal [M, 5]).  ``decoder``
    picks the decoder cell; ``model`` = ops.model_handle(module); ``params`` = list(module.parameters()) (read)."""
    m = _MODELS.get(model)
    if m is None:
        raise RuntimeError('trajnet::lstm_step: unknown model handle (the module was garbage collected)')
    (h2, c2), normal = m.step(m.decoder if decoder else m.encoder, (h, c), obs1, obs2, goals, batch_split,
                              pad_to=(int(pad_to) if pad_to > 0 else None))
    return h2, c2, normal


@lstm_step.register_fake
def _(h, c, obs1, obs2, goals, batch_split, decoder, pad_to, model, params):
    return torch.empty_like(h), torch.empty_like(c), h.new_empty((h.shape[0], 5))


@torch.library.custom_op('trajnet::lstm_sequence', mutates_args=())
def lstm_sequence(observed: torch.Tensor, goals: Optional[torch.Tensor], batch_split: torch.Tensor,
                  truth: Optional[torch.Tensor], t_dec: int, pad_to: int, model: int,
                  params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """LSTM.forward without gradients (reference lstm/lstm.py:170-264) as one dispatcher-visible call: ``truth`` =
    prediction_truth (teacher forcing) or None with ``t_dec = n_predict - 1``; ``pad_to`` <= 0 = this batch's largest
    scene; ``model`` = ops.model_handle(module); ``params`` = list(module.parameters()) (read, never written)."""
    m = _MODELS.get(model)
    if m is None:
        raise RuntimeError('trajnet::lstm_sequence: unknown model handle (the module was garbage collected)')
    rel, pred, _ = m._run_sequence(observed, goals, batch_split, truth, int(t_dec), pad_to=(int(pad_to) if pad_to > 0 else None))
    return rel, pred


@lstm_sequence.register_fake
def _(observed, goals, batch_split, truth, t_dec, pad_to, model, params):
    t_obs, M = observed.shape[0], observed.shape[1]
    S = t_obs - 1 + t_dec
    dev = params[0].device if len(params) else observed.device
    return (torch.empty((S, M, 5), dtype=torch.float32, device=dev),
            torch.empty((S + (1 if t_obs == 2 else 0), M, 2), dtype=torch.float32, device=dev))


# ---- the training sequence: forward with saves + backward sweep behind an opaque handle ---------------------------------
class _SavedSequence(object):
    """what torch.autograd.Function's ctx is to lstm/training.py's SequenceFn: attribute bag + save_for_backward"""
    saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass            # the op's backward is handed its gradients explicitly


_SAVED = collections.OrderedDict()     # handle -> _SavedSequence; bounded: a forward whose backward never runs must not leak
_SAVED_MAX = 16


def set_saved_sequences_limit(n):
    """How many training forward passes may be outstanding (started, backward not yet run) under torch.compile before the
    oldest one's saved state is dropped (default 16; each holds the per-step buffers of its sequence)."""
    global _SAVED_MAX
    if int(n) < 1:
        raise ValueError('limit must be >= 1')
    _SAVED_MAX = int(n)

_next_handle = itertools.count(1)


@torch.library.custom_op('trajnet::lstm_sequence_train', mutates_args=())
def lstm_sequence_train(observed: torch.Tensor, goals: Optional[torch.Tensor], batch_split: torch.Tensor,
                        truth: Optional[torch.Tensor], t_dec: int, pad_to: int, model: int,
                        params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """LSTM.forward in training mode (teacher forcing / free running, reference lstm/lstm.py:170-264) with everything its
    backward sweep needs kept behind ``handle`` (an int64 scalar on the host): (rel_pred, pred, h_last, handle).  Same code
    as the eager autograd.Function (lstm/training.py SequenceFn.forward).  ``params`` = the parameters gradients are wanted
    for: LSTM.forward lists the ones the sequence touches, so the others stay out of the graph and keep ``.grad = None`` as in
    eager mode and in the reference (optimisers with weight decay skip them).  Difference of this path: gradients with
    respect to the observed positions are not formed."""
    m = _MODELS.get(model)
    if m is None:
        raise RuntimeError('trajnet::lstm_sequence_train: unknown model handle (the module was garbage collected)')
    from .lstm.training import SequenceFn, _param_lists
    ctx = _SavedSequence()
    opts = {'pad_to': int(pad_to) if pad_to > 0 else None}
    # `params` is what the caller wants gradients for (LSTM.forward passes the parameters the sequence touches,
    # training.unused_parameter_names() stay out of the graph and keep .grad = None as in eager mode); the sequence itself
    # runs on the module's own parameters
    from .lstm.training import unused_parameter_names
    names, own = _param_lists(m)
    skip = set(unused_parameter_names(m, int(t_dec)))
    used = [n for n in names if n not in skip]
    if len(params) == len(used):            # what LSTM.forward passes (positional: opcheck hands in copies of the tensors)
        ctx.grad_names = used
    elif len(params) == len(names):         # every parameter of the module, in named_parameters() order
        ctx.grad_names = list(names)
    else:
        raise RuntimeError('trajnet::lstm_sequence_train: `params` must be the module\'s parameters in named_parameters() order, '
                           'all %d of them or the %d the sequence touches (got %d)' % (len(names), len(used), len(params)))
    with torch.no_grad():
        rel, pred, h_last = SequenceFn.forward(ctx, m, observed, goals, batch_split, truth, int(t_dec), opts,
                                               *[p.detach() for p in own])
    h = next(_next_handle)
    _SAVED[h] = ctx
    while len(_SAVED) > _SAVED_MAX:
        _SAVED.popitem(last=False)
    return rel, pred, h_last, torch.tensor(h, dtype=torch.int64)


@lstm_sequence_train.register_fake
def _(observed, goals, batch_split, truth, t_dec, pad_to, model, params):
    t_obs, M = observed.shape[0], observed.shape[1]
    S = t_obs - 1 + t_dec
    dev = params[0].device if len(params) else observed.device
    m = _MODELS.get(model)
    H = m.hidden_dim if m is not None else 128
    return (torch.empty((S, M, 5), dtype=torch.float32, device=dev),
            torch.empty((S + (1 if t_obs == 2 else 0), M, 2), dtype=torch.float32, device=dev),
            torch.empty((M, H), dtype=torch.float32, device=dev), torch.empty((), dtype=torch.int64))


@torch.library.custom_op('trajnet::lstm_sequence_backward', mutates_args=())
def lstm_sequence_backward(handle: torch.Tensor, d_rel: torch.Tensor, d_pred: torch.Tensor, d_hlast: torch.Tensor,
                           params: List[torch.Tensor]) -> List[torch.Tensor]:
    """the backward sweep of the sequence ``handle`` names (lstm/training.py SequenceFn.backward): one gradient per entry of
    ``params`` (zeros where the forward did not use the parameter).  A handle serves one backward."""
    ctx = _SAVED.pop(int(handle), None)
    if ctx is None:
        raise RuntimeError('trajnet::lstm_sequence_backward: the saved state of this forward pass is gone (its backward has '
                           'already run, or more than %d forward passes were started since: raise '
                           'trajnetplusplusbaselines_amd.ops.set_saved_sequences_limit())' % _SAVED_MAX)
    from .lstm.training import SequenceFn
    out = SequenceFn.backward(ctx, d_rel, d_pred, d_hlast)
    by_name = dict(zip(ctx.param_names, out[len(out) - len(ctx.param_names):]))
    grads = [by_name.get(n) for n in ctx.grad_names]
    # (a parameter the caller listed although the sequence does not touch it: zeros -- an op cannot return None in a list)
    # ... and its returns may not share storage: the bias gradients are slices of one buffer (lstm/training.py) -- own copies here
    return [(g.clone() if g._base is not None else g) if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]


@lstm_sequence_backward.register_fake
def _(handle, d_rel, d_pred, d_hlast, params):
    return [torch.empty_like(p) for p in params]


def _train_setup_context(ctx, inputs, output):
    ctx.handle = output[3]
    ctx.params = inputs[7]
    ctx.n_inputs = len(inputs)


def _train_backward(ctx, d_rel, d_pred, d_hlast, d_handle):
    grads = torch.ops.trajnet.lstm_sequence_backward(ctx.handle, d_rel, d_pred, d_hlast, ctx.params)
    return (None,) * (ctx.n_inputs - 1) + (grads,)


torch.library.register_autograd('trajnet::lstm_sequence_train', _train_backward, setup_context=_train_setup_context)
