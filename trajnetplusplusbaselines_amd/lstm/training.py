"""Training path of the LSTM forecaster: autograd through the whole sequence (reference lstm/trainer.py:229-269 calls
``loss.backward()`` on the outputs of ``LSTM.forward``).

Forward: ONE ``tnp_lstm_forward_train`` call -- the fused sequence driver of the inference path (encoder and decoder
steps, feedback of the predicted positions), which additionally leaves what the backward pass needs (states, LSTMCell
input ``X``, embedding-MLP activations, post-activation gates, social encodings, winner tables, the positions every
step ran on) in per-step slices of buffers allocated once per sequence.  Nothing is recomputed in the backward pass
except, when the first embedding layer's backward runs dense, the dense grid of a step.

Backward: an explicit reverse sweep.  Per step: ``tnp_h2n_backward`` and ``tnp_lstm_cell_backward`` (pointwise
derivatives from the saved gates), data-gradient GEMMs on the fp32 MFMA kernel against weights transposed once per
sweep, ``tnp_relu_mask`` for the ReLU derivatives, and for social pooling ``tnp_pool_pair_cells`` +
``tnp_social_scatter_backward`` (every in-range neighbour receives the gradient of its cell, overwritten duplicates
included -- SURVEY.md 8a quirk 4).  Weight gradients are contractions over the tracks of ALL steps, so they run as ONE
GEMM per parameter over the stacked per-step operands at the end of the sweep.  Positions fed back to the decoder are
detached exactly as in the reference (lstm/lstm.py:240-250), hidden states pooled from neighbours are not
(lstm/lstm.py:26), so BPTT couples the agents of a scene through the social encoding.
"""
import ctypes
import os

import torch

from .. import _lib

#: The sparse first layer's weight gradient (hit lists -> per-cell contraction -> re-layout) on a side stream beside the grouped
#: weight-gradient launch (they share no operand either WRITES).  OFF: measured slower twice.  At training batches (config 2,
#: 38912 stacked rows) the two matrix-pipe kernels stream 160 MB operands each and evict each other from the L2s (3.74 against
#: 3.565 ms per step, profiles/round5_train_side_stream.txt).  At the trainer's default batch_size 8 (~5900 rows) neither chain
#: fills the chip and the kernels do overlap (~85 us of the step's 1.89 ms) -- but a second active stream makes EVERY launch of
#: the process dearer on the host (tnp_lstm_backward_sweep 0.25 -> 0.375 ms, the forward 0.47 -> 0.55 ms for the same launches),
#: and the step is host-bound there: 1.91 -> 2.06-2.10 ms (round 6, tools/diag/small_batch_workload.py train_time, alternating
#: runs).  TNP_BWD_SIDE_STREAM=1 turns it on (single process only: a gradient reducer orders its collectives against the main
#: stream); the lean form below allocates from the main stream's pool and keeps the buffers until the join.
_SIDE_STREAM = os.environ.get('TNP_BWD_SIDE_STREAM', '0')
_side_streams = {}


def _side_stream(dev):
    """(stream, fork event, join event) of the device, created once"""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = (torch.cuda.Stream(device=dev), torch.cuda.Event(), torch.cuda.Event())
    return st


def _lin(x, w, b=None, relu=False, out=None):
    return _lib.linear_forward(x, w, b, relu=relu, out=out)


def _mm(a, b_t):
    """a [M,K] @ b_t[N,K]^T on the matrix cores."""
    return _lib.linear_forward(a.contiguous(), b_t.contiguous(), None)


def _transpose(x):
    """[R, C] -> contiguous [C, R] with the LDS-tiled kernel (tnp_transpose)."""
    x = x if x.stride(-1) == 1 else x.contiguous()
    out = torch.empty(x.shape[1], x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tnp_transpose(_lib.ptr(x), x.stride(0), x.shape[0], x.shape[1], _lib.ptr(out), x.shape[0],
                                        _lib.stream_ptr()), 'tnp_transpose')
    return out


def _off(t, floats):
    """device pointer `floats` elements past the start of tensor t"""
    return ctypes.c_void_p(t.data_ptr() + 4 * floats)


class WgradProblem(ctypes.Structure):
    """mirror of ``struct tnp_wgrad_problem`` (include/trajnet_hip.h)"""
    _fields_ = [('dy', ctypes.c_void_p), ('ld_dy', ctypes.c_int), ('x', ctypes.c_void_p), ('ld_x', ctypes.c_int),
                ('K', ctypes.c_int), ('Mo', ctypes.c_int), ('No', ctypes.c_int), ('dw', ctypes.c_void_p), ('ld_dw', ctypes.c_int),
                ('dbias', ctypes.c_void_p)]


class TransposeProblem(ctypes.Structure):
    """mirror of ``struct tnp_transpose_problem`` (include/trajnet_hip.h)"""
    _fields_ = [('src', ctypes.c_void_p), ('ld_in', ctypes.c_int), ('rows', ctypes.c_int), ('cols', ctypes.c_int),
                ('out', ctypes.c_void_p), ('ld_out', ctypes.c_int)]


class StepSaves(ctypes.Structure):
    """mirror of ``struct tnp_step_saves`` (include/trajnet_hip.h)"""
    _fields_ = [('X', ctypes.c_void_p), ('act', ctypes.c_void_p * 2), ('gates', ctypes.c_void_p), ('enc', ctypes.c_void_p),
                ('nn_attrs', ctypes.c_void_p), ('winners', ctypes.c_void_p)]


class TrainSaves(ctypes.Structure):
    """mirror of ``struct tnp_train_saves`` (include/trajnet_hip.h)"""
    _fields_ = [('h_all', ctypes.c_void_p), ('c_all', ctypes.c_void_p), ('X_all', ctypes.c_void_p),
                ('act_all', ctypes.c_void_p * 2), ('gates_all', ctypes.c_void_p), ('enc_all', ctypes.c_void_p),
                ('nn_attrs_all', ctypes.c_void_p), ('winners_all', ctypes.c_void_p), ('obs1_all', ctypes.c_void_p),
                ('obs2_all', ctypes.c_void_p), ('h_clean', ctypes.c_void_p), ('ph_all', ctypes.c_void_p),
                ('pc_all', ctypes.c_void_p), ('pgates_all', ctypes.c_void_p), ('traj_in_all', ctypes.c_void_p),
                ('pvec_all', ctypes.c_void_p)]


def _pool_kind(pool):
    """which backward the interaction module needs"""
    name = type(pool).__name__
    if pool is None:
        return 'none'
    return {'NearestNeighborMLP': 'nn', 'HiddenStateMLPPooling': 'hiddenmlp', 'AttentionMLPPooling': 'attention',
            'NearestNeighborLSTM': 'stateful', 'TrajectronPooling': 'stateful'}.get(name, 'grid')


class BwdSweep(ctypes.Structure):
    """mirror of ``struct tnp_bwd_sweep`` (include/trajnet_hip.h)"""
    _fields_ = [('model', ctypes.POINTER(_lib.LstmModel)), ('saves', ctypes.POINTER(TrainSaves)),
                ('S', ctypes.c_int32), ('M', ctypes.c_int32), ('B', ctypes.c_int32), ('n_max', ctypes.c_int32),
                ('n_enc', ctypes.c_int32), ('pos_offset', ctypes.c_int32), ('nn_pool', ctypes.c_int32),
                ('social_sparse', ctypes.c_int32), ('directional_in', ctypes.c_int32), ('h_override_step', ctypes.c_int32),
                ('h_override', ctypes.c_void_p), ('scene_start', ctypes.c_void_p), ('scene_slots', ctypes.c_void_p),
                ('d_rel', ctypes.c_void_p),
                ('d_pred', ctypes.c_void_p), ('wT_enc', ctypes.c_void_p), ('wT_dec', ctypes.c_void_p),
                ('layT', ctypes.c_void_p * 3), ('whT', ctypes.c_void_p), ('w_cell_major', ctypes.c_void_p),
                ('row_base', ctypes.c_void_p), ('row_count', ctypes.c_void_p), ('cells_all', ctypes.c_void_p),
                ('ego_list', ctypes.c_void_p), ('ego_count', ctypes.c_void_p), ('dlin_all', ctypes.c_void_p),
                ('dG_all', ctypes.c_void_p), ('de_all', ctypes.c_void_p), ('dgoal_all', ctypes.c_void_p),
                ('dy_all', ctypes.c_void_p * 3), ('denc_all', ctypes.c_void_p), ('dnn_all', ctypes.c_void_p),
                ('dvel_pool_all', ctypes.c_void_p), ('grid_all', ctypes.c_void_p), ('dh', ctypes.c_void_p),
                ('dc', ctypes.c_void_p), ('hidden_mlp', ctypes.c_int32), ('hm_G_all', ctypes.c_void_p),
                ('hm_R_all', ctypes.c_void_p), ('attention', ctypes.c_int32), ('at_WuT', ctypes.c_void_p),
                ('at_WqT', ctypes.c_void_p), ('at_eself_all', ctypes.c_void_p), ('at_q_all', ctypes.c_void_p),
                ('at_dq_all', ctypes.c_void_p), ('at_ebar_all', ctypes.c_void_p), ('at_du_all', ctypes.c_void_p),
                ('at_A_all', ctypes.c_void_p), ('stateful', ctypes.c_int32), ('st_pwT', ctypes.c_void_p),
                ('st_h2pT', ctypes.c_void_p), ('st_zeros', ctypes.c_void_p), ('st_dG_all', ctypes.c_void_p),
                ('st_dfeat_all', ctypes.c_void_p), ('st_dph', ctypes.c_void_p), ('st_dpc', ctypes.c_void_p),
                ('cellwin_all', ctypes.c_void_p), ('hm_wslot_all', ctypes.c_void_p), ('at_posrec_all', ctypes.c_void_p)]


def _attention_param_grads(pool, P, grads, wgrad, dout_all, bufs, denc_all, h_prev_all, rows, L, dev, sp):
    """Parameter gradients of AttentionMLPPooling from the stacked operands of the sweep.  The kernels work with the
    folded maps (q = Wq_eff e + bq, u = [Wk_eff^T q ; bk . q], out = wop (wo (Win_v (wv ebar) + b_v) + bo) + bop); the
    chain rule back to wq / wk / wv / in_proj / out_proj / out_projection is a handful of [rows, D] GEMMs."""
    ms, mv, mh, D = pool.mlp_dim_spatial, pool.mlp_dim_vel, pool.mlp_dim_hidden, pool.mlp_dim
    g = lambda n: P['pool.' + n].detach()
    wq, wk, wv = g('wq.weight'), g('wk.weight'), g('wv.weight')
    w_in, b_in = g('multihead_attn.in_proj_weight'), g('multihead_attn.in_proj_bias')
    wo, bo = g('multihead_attn.out_proj.weight'), g('multihead_attn.out_proj.bias')
    wop = g('out_projection.weight')
    flat = lambda t: t.reshape(rows, -1)
    eself, q, dq, ebar, du = (flat(bufs[k]) for k in ('eself', 'q', 'dq', 'ebar', 'du'))
    dout = flat(dout_all)

    def wg(dy, x, with_bias):       # (dy^T x, column sums of dy) through the shared split-K kernel
        wgrad('_t', dy, x, '_b' if with_bias else None, immediate=True)
        return grads.pop('_t'), (grads.pop('_b') if with_bias else None)
    # query / key paths
    dwq_eff, dbq = wg(dq, eself, True)                                   # q = Wq_eff e_self + bq
    dwk_eff, _ = wg(q, du[:, :D], False)                                 # u[0:D] = Wk_eff^T q
    dbk, _ = wg(du[:, D:D + 1], q, False)                                # u[D] = bk . q
    # value / output path, recomputed forward intermediates
    t1 = _lib.linear_forward(ebar, wv, None)
    t2 = _lib.linear_forward(t1, w_in[2 * D:].contiguous(), b_in[2 * D:].contiguous())
    t3 = _lib.linear_forward(t2, wo, bo)
    dt3 = _lib.linear_forward(dout, wop.t().contiguous(), None)
    dt2 = _lib.linear_forward(dt3, wo.t().contiguous(), None)
    dt1 = _lib.linear_forward(dt2, w_in[2 * D:].t().contiguous(), None)
    grads['pool.out_projection.weight'], grads['pool.out_projection.bias'] = wg(dout, t3, True)
    grads['pool.multihead_attn.out_proj.weight'], grads['pool.multihead_attn.out_proj.bias'] = wg(dt3, t2, True)
    dwin_v, db_v = wg(dt2, t1, True)
    grads['pool.wv.weight'], _ = wg(dt1, ebar, False)
    # un-fold: Wq_eff = Win_q wq, Wk_eff = Win_k wk
    grads['pool.multihead_attn.in_proj_weight'] = torch.cat([dwq_eff @ wq.t(), dwk_eff @ wk.t(), dwin_v], dim=0)
    grads['pool.multihead_attn.in_proj_bias'] = torch.cat([dbq, dbk.reshape(-1), db_v], dim=0)
    grads['pool.wq.weight'] = w_in[:D].t() @ dwq_eff
    grads['pool.wk.weight'] = w_in[D:2 * D].t() @ dwk_eff
    # embeddings behind the attention
    if mh:
        wgrad('pool.hidden_embedding.0.weight', denc_all, h_prev_all, 'pool.hidden_embedding.0.bias')
    cols = ms + mv
    nb = L.tnp_colsum_prod_workspace_bytes(rows, cols)
    cws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dW2, db2 = torch.empty(cols, 2, device=dev), torch.empty(cols, device=dev)
    _lib.check(L.tnp_colsum_prod(None, _lib.ptr(bufs['A']), rows, cols, _lib.ptr(dW2), _lib.ptr(db2), _lib.ptr(cws), nb, sp()),
               'colsum')
    grads['pool.spatial_embedding.0.weight'], grads['pool.spatial_embedding.0.bias'] = dW2[:ms], db2[:ms]
    if mv:
        grads['pool.vel_embedding.0.weight'], grads['pool.vel_embedding.0.bias'] = dW2[ms:], db2[ms:]


def _nongrid_position_grads(pool, P, S, M, idx, o1_all, o2_all, dnn_all, st_bufs, st_saves, hm_G_all, hm_wslot_all, L, dev, sp,
                            at_posrec_all=None):
    """(d obs1, d obs2) [S, M, 2]: what autograd sends to the positions THROUGH a non-grid interaction module (the frames of an
    S-GAN discriminator carry gradient in a generator step; the reference's trainer gives the discriminator a copy of the
    generator's module, sgan/trainer.py:590-592).  NearestNeighborMLP / NearestNeighborLSTM: through the gathered
    [rel pos | rel vel] attributes of the selected neighbours (the selection itself has no gradient); TrajectronPooling: through
    the ego's [pos, vel] and the whole-batch sum over the other visible tracks; HiddenStateMLPPooling: through the relative
    position / velocity of the slot each max-pooled dimension was routed to; AttentionMLPPooling: through the relative position /
    velocity of every (ego, slot) pair (records left by the sweep's pair kernel)."""
    name = type(pool).__name__
    R = S * M
    d_o1 = torch.empty(S, M, 2, device=dev)
    d_o2 = torch.empty(S, M, 2, device=dev)
    rb_all, rc_all, _ = idx.stacked_rows(S)
    if name in ('NearestNeighborMLP', 'NearestNeighborLSTM'):
        n = pool.n
        in_dim = pool.input_dim if name == 'NearestNeighborMLP' else 4
        W = _lib.f32c(P['pool.embedding.0.weight'].detach(), dev)
        d = W.shape[0]
        dpre = dnn_all if name == 'NearestNeighborMLP' else st_bufs['dfeat']         # gradient of the embedding's pre-activation
        sel = torch.empty(R, n, dtype=torch.int32, device=dev)
        ga = torch.empty(R, n, 4, device=dev)
        _lib.check(L.tnp_pool_nn_pos_backward(_lib.ptr(o1_all), _lib.ptr(o2_all), _lib.ptr(rb_all), _lib.ptr(rc_all), R, idx.n_max, n,
                                              in_dim, _lib.ptr(W), d, _lib.ptr(dpre), n * d, _lib.ptr(sel), _lib.ptr(ga),
                                              _lib.ptr(d_o1), _lib.ptr(d_o2), sp()), 'tnp_pool_nn_pos_backward')
    elif name == 'TrajectronPooling':
        # features = [pos_i, vel_i | sum over the OTHER visible tracks of the whole batch of (pos_j, vel_j)] (:513-529)
        W = P['pool.embedding.0.weight'].detach()                                   # [P, 8]
        din = _mm(st_bufs['dfeat'].reshape(R, -1), W.t()).reshape(S, M, 8)          # dfeat . W on the fp32 matrix cores
        vis = (torch.isfinite(o1_all).all(dim=2) & torch.isfinite(o2_all).all(dim=2)).unsqueeze(2).to(din.dtype)
        own, ssum = din[..., :4] * vis, din[..., 4:] * vis
        d4 = own + (ssum.sum(dim=1, keepdim=True) - ssum) * vis
        d_o2 = d4[..., :2] + d4[..., 2:]
        d_o1 = -d4[..., 2:]
    elif name == 'HiddenStateMLPPooling':
        Ws = _lib.f32c(P['pool.spatial_embedding.0.weight'].detach(), dev)
        Wv = _lib.f32c(P['pool.vel_embedding.0.weight'].detach(), dev) if pool.mlp_dim_vel else Ws
        _lib.check(L.tnp_pool_hiddenmlp_pos_backward(_lib.ptr(hm_G_all), _lib.ptr(hm_wslot_all), _lib.ptr(Ws), _lib.ptr(Wv),
                                                     _lib.ptr(rb_all), _lib.ptr(rc_all), R, M, pool.mlp_dim_spatial, pool.mlp_dim_vel,
                                                     _lib.ptr(d_o1), _lib.ptr(d_o2), sp()), 'tnp_pool_hiddenmlp_pos_backward')
    elif name == 'AttentionMLPPooling':
        # the pair kernel of the sweep left d(rel pos), d(rel vel) per (ego, slot); the ego's own slot cancels (rel = 0)
        _lib.check(L.tnp_pool_pair_pos_gather(_lib.ptr(at_posrec_all), _lib.ptr(rb_all), _lib.ptr(rc_all), R, idx.n_max,
                                              _lib.ptr(d_o1), _lib.ptr(d_o2), sp()), 'tnp_pool_pair_pos_gather')
    else:
        raise NotImplementedError('position gradients through %s' % name)
    return d_o1, d_o2


def _train_saves(h_all, c_all, X_all, gates_all, act_all, enc_all, attrs_all, win_all, o1_all, o2_all, st=None):
    sv = TrainSaves()
    if st is not None and st.get('pvec') is not None:     # pool_to_input=False
        sv.pvec_all = st['pvec'].data_ptr()
    if st is not None and st.get('ph') is not None:    # stateful interaction encoders
        sv.ph_all, sv.pc_all, sv.pgates_all = st['ph'].data_ptr(), st['pc'].data_ptr(), st['pgates'].data_ptr()
        sv.traj_in_all = st['traj_in'].data_ptr() if st.get('traj_in') is not None else None
    sv.h_all, sv.c_all, sv.X_all, sv.gates_all = h_all.data_ptr(), c_all.data_ptr(), X_all.data_ptr(), gates_all.data_ptr()
    for li, a in enumerate(act_all):
        sv.act_all[li] = a.data_ptr()
    sv.enc_all = enc_all.data_ptr() if enc_all is not None else None
    sv.nn_attrs_all = attrs_all.data_ptr() if attrs_all is not None else None
    sv.winners_all = win_all.data_ptr() if win_all is not None else None
    sv.obs1_all, sv.obs2_all = o1_all.data_ptr(), o2_all.data_ptr()
    return sv, (h_all, c_all, X_all, gates_all, act_all, enc_all, attrs_all, win_all, o1_all, o2_all, st)


class SequenceFn(torch.autograd.Function):
    """rel_pred, pred, h_last = SequenceFn.apply(model, observed, goals, batch_split, truth, T_dec, opts, *params)

    ``opts`` (dict): ``noise`` -- the S-GAN generator's noise vector: after the encoder the hidden state becomes
    [ReLU(mlp_decoder_context(h)) | noise] (sgan/sgan.py:200-221); ``input_grad`` -- also return the gradient with
    respect to ``observed`` (the S-GAN discriminator scores positions predicted by the generator): through the velocity
    embedding and, for directional pooling, through the relative velocities stored in the grid.  ``h_last`` is the
    hidden state after the last step (the discriminator's classifier input)."""

    @staticmethod
    def forward(ctx, model, observed, goals, batch_split, truth, T_dec, opts, *params):
        dev = params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('LSTM parameters live on %s: move the model to a ROCm device (model.to("cuda")); '
                               'the MI355X path has no CPU fallback' % dev)
        opts = opts or {}
        # outputs nobody differentiates (the positions and the last hidden state under the trainer's loss) arrive in backward()
        # as None, not as zero tensors autograd filled for them (two fills and a copy per step)
        ctx.set_materialize_grads(False)
        noise = opts.get('noise')
        h_scale = None
        if opts.get('h_scale_arg'):        # VAE: the [M, H] multiplier rides behind the parameters (it needs a gradient)
            h_scale, params = params[-1], params[:-1]
        ctx.has_h_scale = h_scale is not None
        ctx.reduce_fn = opts.get('reduce_fn')
        ctx.input_grad = bool(opts.get('input_grad')) and observed.requires_grad
        if ctx.input_grad and T_dec != 0:
            raise NotImplementedError('gradients with respect to the observed positions: encoder-only runs')
        observed = _lib.f32c(observed.detach(), dev)
        truth = _lib.f32c(truth.detach(), dev) if truth is not None else None
        goals_t = _lib.f32c(goals.detach(), dev) if (goals is not None and model.goal_flag) else None
        T_obs, M = observed.size(0), observed.size(1)
        idx = _lib.SceneIndex.get(batch_split, dev, opts.get('pad_to'))
        H = model.hidden_dim
        pool = model.pool
        S = (T_obs - 1) + T_dec
        L = _lib.lib()
        m, keep, _ = model._descriptor()
        ws, need = model._workspace(m, M, idx.B, dev)
        I = model.encoder.weight_ih.shape[1]
        kind = _pool_kind(pool)
        nn_pool, hm_pool, at_pool, st_pool = kind == 'nn', kind == 'hiddenmlp', kind == 'attention', kind == 'stateful'
        layers = pool.embedding_layers() if kind == 'grid' else []
        if len(layers) > 3:
            raise NotImplementedError('embedding MLPs deeper than three layers')
        # per-step slices of buffers allocated once per sequence
        h_all = torch.empty(S + 1, M, H, device=dev)
        c_all = torch.empty(S + 1, M, H, device=dev)
        X_all = torch.empty(S, M, I, device=dev)
        gates_all = torch.empty(S, M, 4 * H, device=dev)
        act_all = [torch.empty(S, M, lin.weight.shape[0], device=dev) for lin in layers[:-1]]
        enc_all = torch.empty(S, M, pool.pooling_dim, device=dev) if (kind == 'grid' and pool.type_ == 'social') else None
        if hm_pool:    # the max-pooled vector (out_projection's input) and the hidden embedding's pre-activation
            act_all = [torch.empty(S, M, pool.mlp_dim, device=dev)]
        if hm_pool or at_pool:
            enc_all = torch.empty(S, M, pool.mlp_dim_hidden, device=dev) if pool.mlp_dim_hidden else None
        attrs_all = torch.empty(S, M, pool.n * pool.input_dim, device=dev) if nn_pool else None
        st_saves = None
        if st_pool:   # BPTT through the interaction encoder's own LSTM: its features, states and gates of every step
            Hp, nn_lstm = pool.hidden_dim, type(pool).__name__ == 'NearestNeighborLSTM'
            act_all = [torch.empty(S, M, pool.out_dim, device=dev)]
            st_saves = dict(ph=torch.empty(S + 1, M, Hp, device=dev), pc=torch.empty(S + 1, M, Hp, device=dev),
                            pgates=torch.empty(S, M, 4 * Hp, device=dev),
                            traj_in=None if nn_lstm else torch.empty(S, M, 8, device=dev))
            if nn_lstm:
                attrs_all = torch.empty(S, M, pool.n * 4, device=dev)
        # sparse first embedding layer: keep every step's winner table for the sparse backward
        win_all = None
        if kind == 'grid' and pool.type_ == 'social' and opts.get('sparse_backward', getattr(model, 'sparse_backward', True)):
            n1 = layers[0].weight.shape[0]
            lds_ok = n1 * 64 + 4096 <= (160 * 1024) // ((pool.pooling_dim + 15) // 16)    # tnp_social_dgrid_cells' weight block
            if n1 % 64 == 0 and lds_ok and L.tnp_lstm_sparse_first_layer(ctypes.byref(m), M) == 1:
                win_all = torch.empty(S, M, pool.n * pool.n, dtype=torch.int16, device=dev)
        normals = torch.empty(S, M, 5, device=dev)
        o1_all = torch.empty(S, M, 2, device=dev)
        o2_all = torch.empty(S, M, 2, device=dev)
        pos_all = torch.empty(S + (1 if T_obs == 2 else 0), M, 2, device=dev)
        if pool is not None and not model.pool_to_input:   # the interaction vector before it is added to the hidden state
            st_saves = dict(st_saves or {}, pvec=torch.empty(S, M, H, device=dev))
        sv, sv_keep = _train_saves(h_all, c_all, X_all, gates_all, act_all, enc_all, attrs_all, win_all, o1_all, o2_all, st_saves)
        ex = _lib.LstmExtras()
        ctx.noise_at = None
        if noise is not None:        # adding_noise (sgan/sgan.py:200-221): h <- [ReLU(W_ctx h + b_ctx) | z]
            pd = dict(zip([n for n, _ in model.named_parameters()], params))
            wc = _lib.f32c(pd['mlp_decoder_context.0.weight'].detach(), dev)
            bc = _lib.f32c(pd['mlp_decoder_context.0.bias'].detach(), dev)
            z = _lib.f32c(noise, dev)
            nd = int(z.shape[-1])
            groups = z.numel() // nd                 # [k, noise_dim]: k samples batched as k replicas of the scenes
            z = z.reshape(-1)
            h_enc = torch.empty(M, H, device=dev)
            sv.h_clean = h_enc.data_ptr()
            ex.W_ctx, ex.b_ctx, ex.noise, ex.noise_dim = _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(z), nd
            if groups > 1:
                if M % groups:
                    raise ValueError('%d noise vectors for %d tracks' % (groups, M))
                ex.noise_group_tracks = M // groups
        if h_scale is not None:      # VAE add_noise (vae/vae.py:89-107): hidden <- hidden * vae_decoder(z) after the encoder
            if noise is not None:
                raise ValueError('noise and h_scale are exclusive')
            g_scale = _lib.f32c(h_scale.detach(), dev)
            if tuple(g_scale.shape) != (M, H):
                raise ValueError('h_scale must be [M, H] = [%d, %d], got %s' % (M, H, tuple(g_scale.shape)))
            h_enc = torch.empty(M, H, device=dev)
            sv.h_clean = h_enc.data_ptr()
            ex.h_scale = _lib.ptr(g_scale)
        # the whole sequence (encoder + decoder steps, feedback of the predicted positions) is one driver call
        _lib.check(L.tnp_lstm_forward_train(
            ctypes.byref(m), _lib.ptr(observed), T_obs, M, _lib.ptr(goals_t), _lib.ptr(idx.starts), _lib.ptr(idx.primary),
            idx.B, idx.n_max, _lib.ptr(idx.slots), _lib.ptr(truth), T_dec, _lib.ptr(normals), _lib.ptr(pos_all), _lib.ptr(ws), need,
            ctypes.byref(ex), ctypes.byref(sv), _lib.stream_ptr()), 'tnp_lstm_forward_train')
        if noise is not None:
            s_noise = T_obs - 1
            ctx.noise_at = (s_noise, h_enc, h_all[s_noise][:, :H - nd].contiguous())
        ctx.scale_at = None
        if h_scale is not None and T_dec > 0:
            ctx.scale_at = (T_obs - 1, h_enc, g_scale)
        decs = [0] * (T_obs - 1) + [1] * T_dec
        del keep
        ctx.model, ctx.idx, ctx.goals = model, idx, goals_t
        ctx.saved = (h_all, c_all, X_all, gates_all, act_all, enc_all, decs)
        ctx.attrs_all = attrs_all
        ctx.hm_pool, ctx.at_pool, ctx.st_saves = hm_pool, at_pool, st_saves
        ctx.obs_all = (o1_all, o2_all)
        ctx.win_all = win_all
        ctx.w_cell_major = model._cell_major_weight(layers[0].weight, pool) if win_all is not None else None
        ctx.pos_offset = 1 if T_obs == 2 else 0
        ctx.param_names = _param_lists(model)[0]
        ctx.save_for_backward(*params)
        ctx.T_obs = T_obs
        return normals, pos_all, h_all[S].clone()

    @staticmethod
    def backward(ctx, d_rel, d_pred, d_hlast):
        model, idx = ctx.model, ctx.idx
        P = dict(zip(ctx.param_names, ctx.saved_tensors))
        h_all, c_all, X_all, gates_all, act_all, enc_all, decs = ctx.saved
        dev = h_all.device
        S, M, H, E = len(decs), idx.M, model.hidden_dim, model.embedding_dim
        pool = model.pool
        GD = model.goal_dim if model.goal_flag else 0
        L = _lib.lib()
        sp = _lib.stream_ptr
        d_rel = _lib.f32c(d_rel, dev) if d_rel is not None else None
        d_pred = _lib.f32c(d_pred, dev) if d_pred is not None else None
        grads = {}

        # weights of the data-gradient GEMMs, transposed once per sweep ([in, out] rows for the NT kernel)
        tq, tq_keep = [], []

        def T_into(out, name):
            """out [in, out_features] <- P[name]^T through the LDS-tiled transpose kernel (an ATen `.t().contiguous()` is a
            strided elementwise copy: 10-23 us per weight, nine of them per sweep).  Queued: every transpose of the sweep
            goes out in ONE launch (flush_transposes, before the first kernel that reads one)."""
            w = P[name].detach()
            w = w if w.is_contiguous() else w.contiguous()
            tq.append(TransposeProblem(w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], out.data_ptr(), out.stride(0)))
            tq_keep.append(w)
            return out

        def flush_transposes():
            if tq:
                table = (TransposeProblem * len(tq))(*tq)
                _lib.check(L.tnp_transpose_grouped(table, len(tq), sp()), 'tnp_transpose_grouped')
                del tq[:]

        def T(name):
            w = P[name]
            return T_into(torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=dev), name)

        # [W_ih^T ; W_hh^T]: one GEMM per step gives the gradients of the cell's input and of its previous hidden state
        def cell_T(pre):
            w_ih, w_hh = P[pre + '.weight_ih'], P[pre + '.weight_hh']
            out = torch.empty(w_ih.shape[1] + w_hh.shape[1], w_ih.shape[0], dtype=torch.float32, device=dev)
            T_into(out[:w_ih.shape[1]], pre + '.weight_ih')
            T_into(out[w_ih.shape[1]:], pre + '.weight_hh')
            return out
        wT = {pre: cell_T(pre) for pre in set('decoder' if d else 'encoder' for d in decs)}
        has_h2n = 'hidden2normal.linear.weight' in P            # the S-GAN discriminator has no output head
        o1_all, o2_all = ctx.obs_all
        st_saves = ctx.st_saves                                  # NearestNeighborLSTM / TrajectronPooling
        st_pool = st_saves is not None and st_saves.get('ph') is not None
        nn_pool = ctx.attrs_all is not None and not st_pool      # NearestNeighborMLP: only its embedding has parameters
        hm_pool, at_pool = ctx.hm_pool, ctx.at_pool              # HiddenStateMLPPooling, AttentionMLPPooling
        grid_pool = pool is not None and not (nn_pool or hm_pool or at_pool or st_pool)
        layers = pool.embedding_layers() if grid_pool else []
        lay_names = ['pool.embedding.%d' % i for i, mod in enumerate(pool.embedding) if isinstance(mod, torch.nn.Linear)] \
            if grid_pool else []
        social = grid_pool and pool.type_ == 'social'
        dnn_all = torch.empty(S, M, pool.out_dim, device=dev) if nn_pool else None
        directional_in = ctx.input_grad and grid_pool and pool.type_ == 'directional'
        pos_grad_pool = ctx.input_grad and (nn_pool or hm_pool or at_pool or st_pool)   # position gradients through a non-grid module
        sparse_bwd = ctx.win_all is not None      # first layer's gradients from the winner tables (csrc/lstm_bwd.hip)
        layT = [T(n + '.weight') if (li > 0 or ((social or directional_in) and not sparse_bwd)) else None
                for li, n in enumerate(lay_names)]
        whT = T('pool.hidden_dim_encoding.weight') if social else None
        hm_G_all = hm_R_all = None
        if hm_pool:
            ms, mv, mh = pool.mlp_dim_spatial, pool.mlp_dim_vel, pool.mlp_dim_hidden
            layT = [T('pool.out_projection.weight')]
            whT = T('pool.hidden_embedding.0.weight') if mh else None
            hm_G_all = torch.empty(S, M, ms + mv, device=dev)
            hm_R_all = torch.empty(S, M, ms + mv, 2, device=dev)

        st_bufs = None
        if st_pool:
            Hp = pool.hidden_dim
            st_bufs = dict(pwT=cell_T('pool.pool_lstm'),
                           h2pT=T('pool.hidden2pool.weight'), zeros=torch.zeros(M, 2, device=dev),
                           dG=torch.empty(S, M, 4 * Hp, device=dev), dfeat=torch.empty(S, M, pool.out_dim, device=dev),
                           dph=torch.zeros(M, Hp, device=dev), dpc=torch.zeros(M, Hp, device=dev))
        at_bufs = None
        if at_pool:   # linear maps around the softmax folded as in the forward (AttentionMLPPooling.folded)
            ms, mv, mh, D = pool.mlp_dim_spatial, pool.mlp_dim_vel, pool.mlp_dim_hidden, pool.mlp_dim
            wq_eff, _, wu_f, wfin, _ = pool.folded()
            layT = [wfin.t().contiguous()]
            whT = T('pool.hidden_embedding.0.weight') if mh else None
            at_wuT, at_wqT = wu_f.t().contiguous(), wq_eff.t().contiguous()
            at_bufs = dict(eself=torch.empty(S, M, D, device=dev), q=torch.empty(S, M, D, device=dev),
                           dq=torch.empty(S, M, D, device=dev), ebar=torch.empty(S, M, D, device=dev),
                           du=torch.empty(S, M, D + 4, device=dev), A=torch.empty(S, M, ms + mv, 3, device=dev))

        # per-step operands of the deferred weight-gradient GEMMs
        dlin_all = torch.empty(S, M, 5, device=dev)
        dG_all = torch.empty(S, M, 4 * H, device=dev)
        de_all = torch.empty(S, M, E - 2, device=dev)
        dgoal_all = torch.empty(S, M, GD - 2, device=dev) if GD else None
        gdir_all = None
        dy_all = [torch.empty(S, M, lin.weight.shape[0], device=dev) for lin in layers]
        denc_all = torch.empty(S, M, pool.pooling_dim, device=dev) if social else None
        grid_all = None
        if st_pool:
            dy_all = [torch.empty(S, M, pool.out_dim, device=dev)]          # gradient of the interaction vector
        if hm_pool or at_pool:
            dy_all = [torch.empty(S, M, pool.out_dim, device=dev)]
            denc_all = torch.empty(S, M, mh, device=dev) if mh else None
            row_base, row_count = idx.row_base, idx.row_count
        if grid_pool:
            G, cell, half_x, half_y = pool._geometry()
            C = pool.pooling_dim
            grid_all = torch.empty(S, M, C * G * G, device=dev) if not sparse_bwd else None
            cells_all = cellwin_all = None
            if social or directional_in:
                row_base, row_count = idx.row_base, idx.row_count
                # pair cells of every step in one launch, AS AUTOGRAD SEES THEM (tnp_pool_pair_cells_autograd): every in-range
                # neighbour receives its cell's gradient (duplicates included), except in a cell whose final value is the
                # constant 0 -- cell (0, 0) clobbered by an out-of-range / absent / padded neighbour: lp_pool2d(x, 1, 1) has a
                # zero derivative there (reference gridbased_pooling.py:304)
                R, ncell = S * M, G * G
                rb_all, rc_all, rp_all = idx.stacked_rows(S)
                cells_raw = torch.empty(S, M, idx.n_max, dtype=torch.int32, device=dev)
                cells_all = torch.empty(S, M, idx.n_max, dtype=torch.int32, device=dev)
                cellwin_all = torch.empty(S, M, idx.n_max, dtype=torch.int32, device=dev) if directional_in else None
                _lib.check(L.tnp_pool_pair_cells_autograd(_lib.ptr(o2_all), _lib.ptr(rb_all), _lib.ptr(rc_all), _lib.ptr(rp_all),
                                                          idx.n_max, R, idx.n_max, G, cell, half_x, half_y, float(pool.constant),
                                                          _lib.ptr(cells_raw), _lib.ptr(cells_all), _lib.ptr(cellwin_all), sp()),
                           'pair_cells_autograd')
                del cells_raw
                if sparse_bwd:
                    # per (step, cell) the egos with a neighbour in that cell
                    occ = torch.empty(R, ncell, dtype=torch.uint8, device=dev)
                    occ_t = torch.empty(ncell, R, dtype=torch.int32, device=dev)
                    ego_list = torch.empty(ncell, R, 2, dtype=torch.int32, device=dev)
                    ego_count = torch.empty(ncell, S, dtype=torch.int32, device=dev)
                    _lib.check(L.tnp_pair_ego_lists(_lib.ptr(cells_all), R, M, idx.n_max, ncell, _lib.ptr(occ), _lib.ptr(occ_t),
                                                    _lib.ptr(ego_list), _lib.ptr(ego_count), sp()), 'ego_lists')
                    del occ, occ_t

        if d_hlast is not None:
            dh, dc = _lib.f32c(d_hlast, dev).clone(), torch.zeros(M, H, device=dev)
        else:
            dh, dc = torch.zeros(2, M, H, device=dev).unbind(0)          # one fill
        dvel_pool_all = torch.empty(S, M, 2, device=dev) if directional_in else None

        # ---- the reverse sweep: one driver call (two around the S-GAN noise hook), csrc/lstm_bwd.hip ----
        sv, sv_keep = _train_saves(h_all, c_all, X_all, gates_all, act_all, enc_all, ctx.attrs_all, ctx.win_all, o1_all, o2_all,
                                   st_saves)
        m, keep, _ = model._descriptor()
        if not has_h2n:
            m.Wn, m.bn = None, None
        n_enc = sum(1 for d in decs if not d)
        sw = BwdSweep()
        sw.model, sw.saves = ctypes.pointer(m), ctypes.pointer(sv)
        sw.S, sw.M, sw.B, sw.n_max, sw.n_enc, sw.pos_offset = S, M, idx.B, idx.n_max, n_enc, ctx.pos_offset
        sw.nn_pool, sw.social_sparse, sw.directional_in = int(nn_pool), int(social and sparse_bwd), int(directional_in)
        sw.h_override_step = -1
        hook_at = ctx.noise_at if ctx.noise_at is not None else ctx.scale_at
        if hook_at is not None:
            sw.h_override_step, sw.h_override = hook_at[0] - 1, hook_at[1].data_ptr()
        sw.scene_start = idx.starts.data_ptr()
        sw.scene_slots = idx.slots.data_ptr() if idx.slots is not None else None
        sw.d_rel = d_rel.data_ptr() if d_rel is not None else None
        sw.d_pred = d_pred.data_ptr() if d_pred is not None else None
        sw.wT_enc = wT['encoder'].data_ptr() if 'encoder' in wT else None
        sw.wT_dec = wT['decoder'].data_ptr() if 'decoder' in wT else None
        for li, t in enumerate(layT):
            sw.layT[li] = t.data_ptr() if t is not None else None
        sw.whT = whT.data_ptr() if whT is not None else None
        sw.w_cell_major = ctx.w_cell_major.data_ptr() if sparse_bwd else None
        if social or directional_in or hm_pool or at_pool:
            sw.row_base, sw.row_count = row_base.data_ptr(), row_count.data_ptr()
        if st_pool:
            sw.stateful = 1
            sw.st_pwT, sw.st_h2pT, sw.st_zeros = (st_bufs[k].data_ptr() for k in ('pwT', 'h2pT', 'zeros'))
            sw.st_dG_all, sw.st_dfeat_all, sw.st_dph, sw.st_dpc = (st_bufs[k].data_ptr() for k in ('dG', 'dfeat', 'dph', 'dpc'))
        at_posrec_all = None
        if at_pool:
            if pos_grad_pool:
                at_posrec_all = torch.empty(S, M, idx.n_max, 4, device=dev)
                sw.at_posrec_all = at_posrec_all.data_ptr()
            sw.attention, sw.at_WuT, sw.at_WqT = 1, at_wuT.data_ptr(), at_wqT.data_ptr()
            sw.at_eself_all, sw.at_q_all, sw.at_dq_all = (at_bufs[k].data_ptr() for k in ('eself', 'q', 'dq'))
            sw.at_ebar_all, sw.at_du_all, sw.at_A_all = (at_bufs[k].data_ptr() for k in ('ebar', 'du', 'A'))
        hm_wslot_all = None
        if hm_pool:
            sw.hidden_mlp, sw.hm_G_all, sw.hm_R_all = 1, hm_G_all.data_ptr(), hm_R_all.data_ptr()
            if pos_grad_pool:
                hm_wslot_all = torch.empty(S, M, pool.mlp_dim_spatial + pool.mlp_dim_vel, dtype=torch.int32, device=dev)
                sw.hm_wslot_all = hm_wslot_all.data_ptr()
        if grid_pool and cells_all is not None:
            sw.cells_all = cells_all.data_ptr()
            sw.cellwin_all = cellwin_all.data_ptr() if cellwin_all is not None else None
        if social and sparse_bwd:
            sw.ego_list, sw.ego_count = ego_list.data_ptr(), ego_count.data_ptr()
        sw.dlin_all, sw.dG_all, sw.de_all = dlin_all.data_ptr(), dG_all.data_ptr(), de_all.data_ptr()
        sw.dgoal_all = dgoal_all.data_ptr() if GD else None
        for li, t in enumerate(dy_all):
            sw.dy_all[li] = t.data_ptr()
        sw.denc_all = denc_all.data_ptr() if denc_all is not None else None
        sw.dnn_all = dnn_all.data_ptr() if nn_pool else None
        sw.dvel_pool_all = dvel_pool_all.data_ptr() if directional_in else None
        sw.grid_all = grid_all.data_ptr() if grid_all is not None else None
        sw.dh, sw.dc = dh.data_ptr(), dc.data_ptr()
        need = L.tnp_lstm_backward_scratch_bytes(ctypes.byref(sw))
        scratch = torch.empty(need, dtype=torch.uint8, device=dev)

        flush_transposes()             # every transposed weight of the sweep: one launch

        def sweep(hi, lo):
            if hi >= lo:
                _lib.check(L.tnp_lstm_backward_sweep(ctypes.byref(sw), hi, lo, _lib.ptr(scratch), need, sp()), 'backward_sweep')

        d_h_scale = None
        if ctx.scale_at is not None:
            # backward of the VAE hook h_dec = h_enc * g: d g = dh * h_enc, d h_enc = dh * g (the cell state passes through)
            s_hook, h_enc, g_scale = ctx.scale_at
            sweep(S - 1, s_hook)
            d_h_scale = dh * h_enc
            dh.mul_(g_scale)
            sweep(s_hook - 1, 0)
        elif ctx.noise_at is None:
            sweep(S - 1, 0)
        else:
            s_noise, h_enc, ctx_act = ctx.noise_at
            sweep(S - 1, s_noise)
            # backward of adding_noise: dh is the gradient of [ReLU(W_ctx h + b_ctx) | z]
            nc = ctx_act.shape[1]
            dctx = torch.empty(M, nc, device=dev)
            _lib.check(L.tnp_relu_mask(_lib.ptr(dh), H, _lib.ptr(ctx_act), nc, M, nc, _lib.ptr(dctx), nc, sp()), 'relu_mask')
            grads['mlp_decoder_context.0.weight'] = _mm(dctx.t(), h_enc.t())
            grads['mlp_decoder_context.0.bias'] = dctx.sum(0)
            w_ctx_T = T('mlp_decoder_context.0.weight')
            flush_transposes()
            dh.copy_(_lin(dctx, w_ctx_T))
            sweep(s_noise - 1, 0)
        del keep, sv_keep
        if GD:    # direction to the goal, the goal embedding's input (all steps at once)
            gd = o2_all - ctx.goals[None]
            nf = gd.norm(dim=2, keepdim=True)
            gdir_all = torch.nan_to_num(torch.where(nf == 0, torch.zeros_like(gd), gd / nf)) * 4.0
        d_obs = None
        if ctx.input_grad:   # vel = o2 - o1 feeds the input embedding (x4) and the directional grid values
            emb_wT4 = torch.zeros(4, E - 2, device=dev)          # [W_emb^T ; 0 0] so that the GEMM's N is a multiple of 4
            emb_wT4[:2] = P['input_embedding.input_embeddings.0.weight'].detach().t()
            dvel = _lin(de_all.reshape(S * M, E - 2), emb_wT4)[:, :2].reshape(S, M, 2) * 4.0
            if directional_in:
                dvel = dvel + dvel_pool_all
            d_obs = torch.zeros(ctx.T_obs, M, 2, device=dev)
            d_obs[1:S + 1] += dvel
            d_obs[0:S] -= dvel
            if pos_grad_pool:
                # the interaction module sees the positions as well (obs1 = frame s, obs2 = frame s + 1 of step s)
                d_o1, d_o2 = _nongrid_position_grads(pool, P, S, M, idx, o1_all, o2_all, dnn_all, st_bufs if st_pool else None,
                                                     st_saves, hm_G_all, hm_wslot_all, L, dev, sp, at_posrec_all)
                d_obs[1:S + 1] += d_o2
                d_obs[0:S] += d_o1
            if GD:   # the goal embedding sees the unit vector from the goal to the current position (lstm/lstm.py:132-139)
                goal_wT4 = torch.zeros(4, GD - 2, device=dev)
                goal_wT4[:2] = P['goal_embedding.input_embeddings.0.weight'].detach().t()
                du = _lin(dgoal_all.reshape(S * M, GD - 2), goal_wT4)[:, :2].reshape(S, M, 2) * 4.0
                unit = gdir_all / 4.0
                d_o2 = (du - unit * (unit * du).sum(dim=2, keepdim=True)) / nf
                d_obs[1:S + 1] += torch.where(torch.isfinite(d_o2) & (nf > 0), d_o2, torch.zeros_like(d_o2))

        # ---- deferred weight gradients: one GEMM per parameter over the stacked steps ----
        # Data-parallel training hands in `reduce_fn` (parallel.GradReducer): every gradient tensor is all-reduced over
        # RCCL as soon as its kernel is enqueued -- asynchronously, on RCCL's stream, largest message first -- so the
        # collectives overlap the remaining weight-gradient GEMMs; the compute stream waits for them before the gradients
        # are handed to autograd.
        reduce_fn = ctx.reduce_fn
        if hasattr(reduce_fn, 'begin'):
            reduce_fn.begin()          # nothing a failed earlier pass left behind travels with this step's message
        pending = []

        pending_tensors = []

        def publish(t):
            if reduce_fn is not None:
                pending_tensors.append(t)
                pending.append(reduce_fn(t))

        wg_ws = [None]
        wg_queue = []        # contractions waiting for the grouped launch: (name, bias_name, problem, tensors kept alive)
        # Bias gradients are slices of ONE buffer: an LSTMCell's bias_hh gradient equals its bias_ih gradient, and one copy of
        # the buffer serves every such pair of the step (a clone launch per pair before)
        bias_arena = [torch.empty(8192, device=dev), 0, {}]        # buffer, floats used, name -> (offset, length)

        def bias_vector(name, n):
            buf, used, where = bias_arena
            if used + n > buf.numel():
                return torch.empty(n, device=dev)
            where[name] = (used, n)
            bias_arena[1] = (used + n + 63) // 64 * 64
            return buf[used:used + n]

        def wgrad_bias(bias_name, dy):
            # the column sums alone (a layer whose weight gradient is formed elsewhere): a contraction without a weight part
            # in the grouped launch (csrc/gemm_wgrad.hip: wgrad_bias_block)
            dy2 = dy.reshape(-1, dy.shape[-1])
            db = bias_vector(bias_name, dy2.shape[1])
            grads[bias_name] = db
            wg_queue.append((None, bias_name, WgradProblem(dy2.data_ptr(), dy2.stride(0), None, 0, dy2.shape[0], dy2.shape[1], 0,
                                                           None, 0, db.data_ptr()), (dy2, None, None, db)))

        def wgrad(name, dy, x, bias_name, immediate=False):
            # dW = dy^T x over the stacked steps, operands as stored, K split across workgroups (csrc/gemm_wgrad.hip).
            # Queued: all contractions of the step go to the device in ONE launch (tnp_wgrad_grouped) when flush_wgrads() runs;
            # immediate: launched now (the attention un-folding consumes its intermediates right away).
            dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
            K, Mo, No = dy2.shape[0], dy2.shape[1], x2.shape[1]
            dw = torch.empty(Mo, No, device=dev)
            db = None
            if bias_name is not None:     # (immediate launches: intermediates the caller combines and publishes itself -- own storage)
                db = torch.empty(Mo, device=dev) if immediate else bias_vector(bias_name, Mo)
            grads[name] = dw
            if bias_name is not None:
                grads[bias_name] = db
            if not immediate:
                wg_queue.append((name, bias_name, WgradProblem(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), K, Mo, No,
                                                               dw.data_ptr(), No, db.data_ptr() if db is not None else None),
                                 (dy2, x2, dw, db)))
                return
            nbytes = L.tnp_wgrad_workspace_bytes(Mo, No, K)
            if wg_ws[0] is None or wg_ws[0].numel() < nbytes:
                wg_ws[0] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.tnp_wgrad(_lib.ptr(dy2), dy2.stride(0), _lib.ptr(x2), x2.stride(0), K, Mo, No, _lib.ptr(dw), No,
                                   _lib.ptr(db), _lib.ptr(wg_ws[0]), nbytes, sp()), 'tnp_wgrad')
            if not name.startswith('_'):       # '_t' / '_b': intermediates of the attention un-folding, combined further on the
                publish(dw)                    # compute stream -- their final forms are published at the end
                if bias_name is not None:
                    publish(db)

        def flush_wgrads():
            if not wg_queue:
                return
            table = (WgradProblem * len(wg_queue))(*[q[2] for q in wg_queue])
            nbytes = L.tnp_wgrad_grouped_workspace_bytes(table, len(wg_queue))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.tnp_wgrad_grouped(table, len(wg_queue), _lib.ptr(ws), nbytes, sp()), 'tnp_wgrad_grouped')
            for name, bias_name, _, keep in wg_queue:
                if keep[2] is not None:
                    publish(keep[2])
                if bias_name is not None:
                    publish(keep[3])
            del wg_queue[:]

        def layer_wgrad(li, name, sp=sp, keep_alive=None):
            if li == 0 and sparse_bwd:
                # dW'[cell][ch][:] from the per-cell hit lists of the whole sweep, then back to the parameter's layout
                N1 = dy_all[0].shape[2]
                hit_t = torch.empty(ncell, R, dtype=torch.int32, device=dev)
                hits = torch.empty(ncell, R, 2, dtype=torch.int32, device=dev)
                count = torch.empty(ncell, dtype=torch.int32, device=dev)
                _lib.check(L.tnp_sparse_hits_build(_lib.ptr(ctx.win_all), _lib.ptr(row_base), R, M, ncell, _lib.ptr(hit_t),
                                                   _lib.ptr(hits), _lib.ptr(count), sp()), 'hits_build')
                dwc = torch.empty(ncell, C, N1, device=dev)
                _lib.check(L.tnp_sparse_wgrad(_lib.ptr(dy_all[0]), N1, _lib.ptr(enc_all), C, _lib.ptr(hits), _lib.ptr(count),
                                              R, C, ncell, N1, _lib.ptr(dwc), sp()), 'sparse_wgrad')
                # back to the parameter's layout: g[n][ch ncell + cell] = dW'[cell][ch][n] is one strided transpose per
                # channel ([ncell, N1] rows C N1 apart -> [N1, ncell] columns of g), all of them in one launch (an ATen
                # permute + reshape copy: 22 us)
                gw = torch.empty(N1, C * ncell, device=dev)
                table = (TransposeProblem * C)(*[
                    TransposeProblem(dwc.data_ptr() + 4 * ch * N1, C * N1, ncell, N1, gw.data_ptr() + 4 * ch * ncell, C * ncell)
                    for ch in range(C)])
                _lib.check(L.tnp_transpose_grouped(table, C, sp()), 'tnp_transpose_grouped')
                grads[name + '.weight'] = gw
                publish(grads[name + '.weight'])
                wgrad_bias(name + '.bias', dy_all[0])          # column sums: ride along in the grouped launch (main stream)
                if keep_alive is not None:     # side stream: nothing these kernels touch goes back to the allocator before the join
                    keep_alive += [hit_t, hits, count, dwc, table]
                return
            wgrad(name + '.weight', dy_all[li], grid_all if li == 0 else act_all[li - 1], name + '.bias')
        side_done, side_keep = None, []
        if lay_names:
            # first: the largest gradient (16.8 MB at config 2) gets the longest overlap with the collectives.  Opt-in (see
            # _SIDE_STREAM: measured slower): on a side stream beside the grouped weight-gradient launch.  Buffers come from THIS
            # stream's allocator (a side stream's own pool meets a new size every ragged batch: a hipMalloc per step) and are
            # kept until the join; the kernels get the side stream's handle.
            side = None
            if sparse_bwd and dev.type == 'cuda' and reduce_fn is None and _SIDE_STREAM not in ('0', ''):
                side, fork, side_done = _side_stream(dev)
                fork.record(torch.cuda.current_stream(dev))
                side.wait_event(fork)                 # everything the sweep left behind is complete for the side stream
                handle = ctypes.c_void_p(side.cuda_stream)
                layer_wgrad(0, lay_names[0], sp=lambda: handle, keep_alive=side_keep)
                side_done.record(side)
            else:
                layer_wgrad(0, lay_names[0])
        h_out_all, h_prev_all = h_all[1:], h_all[:-1]
        if hook_at is not None:     # the last encoder step's output is the hidden state BEFORE the noise was added / the scaling
            h_out_all = h_out_all.clone()
            h_out_all[hook_at[0] - 1] = hook_at[1]
        if has_h2n:
            wgrad('hidden2normal.linear.weight', dlin_all, h_out_all, 'hidden2normal.linear.bias')
        n_enc = sum(1 for d in decs if not d)
        # hidden operand of the LSTMCell: h, or h + interaction vector for LSTM(pool_to_input=False)
        hid_all = h_prev_all
        if ctx.st_saves is not None and ctx.st_saves.get('pvec') is not None:
            hid_all = torch.nan_to_num(h_prev_all + ctx.st_saves['pvec'])
        hh_clones = []
        for pre, lo, hi in (('encoder', 0, n_enc), ('decoder', n_enc, S)):
            if hi > lo:
                wgrad(pre + '.weight_ih', dG_all[lo:hi], X_all[lo:hi], pre + '.bias_ih')
                wgrad(pre + '.weight_hh', dG_all[lo:hi], hid_all[lo:hi], None)
                hh_clones.append((pre + '.bias_hh', pre + '.bias_ih'))
        vel_all = torch.empty_like(o2_all)            # nan_to_num(o2 - o1) * 4: the input embedding's operand, one launch
        _lib.check(L.tnp_scaled_diff(_lib.ptr(o2_all), _lib.ptr(o1_all), o2_all.numel(), 4.0, _lib.ptr(vel_all), sp()), 'scaled_diff')
        wgrad('input_embedding.input_embeddings.0.weight', de_all, vel_all, 'input_embedding.input_embeddings.0.bias')
        if GD:
            wgrad('goal_embedding.input_embeddings.0.weight', dgoal_all, gdir_all, 'goal_embedding.input_embeddings.0.bias')
        for li, name in enumerate(lay_names):
            if li > 0:
                layer_wgrad(li, name)
        if st_pool:
            ph_all = st_saves['ph']
            wgrad('pool.hidden2pool.weight', dy_all[0], ph_all[1:], 'pool.hidden2pool.bias')
            wgrad('pool.pool_lstm.weight_ih', st_bufs['dG'], act_all[0], 'pool.pool_lstm.bias_ih')
            wgrad('pool.pool_lstm.weight_hh', st_bufs['dG'], ph_all[:-1], None)
            hh_clones.append(('pool.pool_lstm.bias_hh', 'pool.pool_lstm.bias_ih'))
            if st_saves['traj_in'] is None:     # NearestNeighborLSTM: rows = (step, track, neighbour slot)
                d = pool.out_dim // pool.n
                wgrad('pool.embedding.0.weight', st_bufs['dfeat'].reshape(-1, d), ctx.attrs_all.reshape(-1, 4), 'pool.embedding.0.bias')
            else:
                wgrad('pool.embedding.0.weight', st_bufs['dfeat'], st_saves['traj_in'], 'pool.embedding.0.bias')
        if at_pool:
            _attention_param_grads(pool, P, grads, wgrad, dy_all[0], at_bufs, denc_all, h_prev_all, S * M, L, dev, sp)
        if hm_pool:
            wgrad('pool.out_projection.weight', dy_all[0], act_all[0], 'pool.out_projection.bias')
            if mh:
                wgrad('pool.hidden_embedding.0.weight', denc_all, h_prev_all, 'pool.hidden_embedding.0.bias')
            # Linear(2 -> dim) embeddings behind the max-pool: per output unit the input is the winning slot's
            rows, cols = S * M, ms + mv
            nb = L.tnp_colsum_prod_workspace_bytes(rows, cols)
            cws = torch.empty(nb, dtype=torch.uint8, device=dev)
            dW2, db2 = torch.empty(cols, 2, device=dev), torch.empty(cols, device=dev)
            _lib.check(L.tnp_colsum_prod(_lib.ptr(hm_G_all), _lib.ptr(hm_R_all), rows, cols, _lib.ptr(dW2), _lib.ptr(db2),
                                         _lib.ptr(cws), nb, sp()), 'colsum_prod')
            grads['pool.spatial_embedding.0.weight'], grads['pool.spatial_embedding.0.bias'] = dW2[:ms], db2[:ms]
            if mv:
                grads['pool.vel_embedding.0.weight'], grads['pool.vel_embedding.0.bias'] = dW2[ms:], db2[ms:]
        if nn_pool:       # rows = (step, track, neighbour slot)
            d = pool.out_dim // pool.n
            wgrad('pool.embedding.0.weight', dnn_all.reshape(-1, d), ctx.attrs_all.reshape(-1, pool.input_dim),
                  'pool.embedding.0.bias')
        if social:
            wgrad('pool.hidden_dim_encoding.weight', denc_all, h_prev_all, 'pool.hidden_dim_encoding.bias')

        flush_wgrads()
        if reduce_fn is not None:      # gradients formed outside wgrad() / layer_wgrad() (torch expressions) are reduced here
            done = set(id(t) for t in pending_tensors)
            done.update(id(t._base) for t in pending_tensors if t._base is not None)     # bias vectors: slices of one buffer,
                                                                                         # each already reduced on its own
            for n in ctx.param_names:
                g = grads.get(n)
                if g is None:
                    continue
                base = g._base if g._base is not None else g      # slices / re-layouts of a buffer: reduce the buffer once
                if id(base) not in done:
                    if not base.is_contiguous():
                        base = base.contiguous()
                        grads[n] = base
                    done.add(id(base))
                    publish(base)
            if hasattr(reduce_fn, 'flush'):        # the small gradients travel as ONE flattened message (GradReducer)
                w = reduce_fn.flush()
                if w is not None:
                    pending.append(w)
            for w in pending:
                w.wait()                  # the compute stream waits for RCCL's stream; no host synchronisation
        if side_done is not None:
            torch.cuda.current_stream(dev).wait_event(side_done)     # join: the first layer's gradient is final for what comes next
            del side_keep[:]
        if hh_clones:
            where = bias_arena[2]
            if all(src in where for _, src in hh_clones):
                twin = bias_arena[0][:bias_arena[1]].clone()
                for dst_name, src_name in hh_clones:
                    off, n = where[src_name]
                    grads[dst_name] = twin[off:off + n]
            else:
                for dst_name, src_name in hh_clones:
                    grads[dst_name] = grads[src_name].clone()
        # parameters the forward never touches get no gradient (None, as autograd does for the reference), so that
        # optimizers skip them: a zero gradient would still let Adam + weight decay move them
        out = [None, d_obs, None, None, None, None, None]
        for n in ctx.param_names:
            out.append(grads.get(n))
        if ctx.has_h_scale:
            out.append(d_h_scale if d_h_scale is not None else torch.zeros(M, H, device=dev))
        return tuple(out)


def unused_parameter_names(model, T_dec):
    """Names of the parameters LSTM.forward never touches for this module configuration and call: the reference's autograd
    leaves their ``.grad`` None (so Adam + weight decay does not move them), and so does SequenceFn.backward.  Static
    rules, checked against what the backward sweep actually returns in tests/test_gpu_training.py:
      goal_embedding.*                       unless goal_flag (lstm/lstm.py:132-139)
      decoder.*                              in an encoder-only run (T_dec == 0: the S-GAN discriminator)
      pool.pool_lstm.*, pool.hidden2pool.*   of a grid module with embedding_arch='lstm_layer' (its forward is
                                             Linear + ReLU, lstm/gridbased_pooling.py:94-110 never calls lstm_forward)
      mlp_decoder_context.*                  (S-GAN generator) only used with a noise vector -- not on the LSTM.forward path"""
    unused = []
    grid_lstm_layer = hasattr(model.pool, 'embedding_layers') and getattr(model.pool, 'embedding_arch', None) == 'lstm_layer'
    for n, _ in model.named_parameters():
        if n.startswith('goal_embedding.') and not model.goal_flag:
            unused.append(n)
        elif n.startswith('decoder.') and T_dec == 0:
            unused.append(n)
        elif grid_lstm_layer and (n.startswith('pool.pool_lstm.') or n.startswith('pool.hidden2pool.')):
            unused.append(n)
        elif n.startswith('mlp_decoder_context.'):
            unused.append(n)
    return unused


def _param_lists(model):
    """(names, parameters) of the model in named_parameters() order (kept by our LSTM between calls)"""
    if hasattr(model, '_named_parameter_lists'):
        return model._named_parameter_lists()
    named = list(model.named_parameters())
    return [n for n, _ in named], [q for _, q in named]


def run_sequence_with_grad(model, observed, goals, batch_split, truth, T_dec, opts=None):
    """(rel_pred, pred, h_last) attached to the autograd graph of the model's parameters (and of `observed` when
    opts['input_grad'] is set and it requires grad)."""
    pool = getattr(model, 'pool', None)
    if getattr(pool, 'pool_size', 1) != 1 or getattr(pool, 'blur_size', 1) != 1:
        raise NotImplementedError('training through GridBasedPooling(pool_size != 1 or blur_size != 1) is not available on the MI355X '
                                  'path (the trainer never sets them); inference (model.eval() / torch.no_grad()) is')
    params = _param_lists(model)[1]
    if opts is not None and opts.get('h_scale') is not None:      # VAE: a differentiable [M, H] multiplier of the encoder's state
        opts = dict(opts, h_scale_arg=True)
        return SequenceFn.apply(model, observed, goals, batch_split, truth, T_dec, opts, *params, opts['h_scale'])
    return SequenceFn.apply(model, observed, goals, batch_split, truth, T_dec, opts, *params)
