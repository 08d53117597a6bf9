"""Training path of the LSTM forecaster: autograd through the whole sequence (reference lstm/trainer.py:229-269 calls
``loss.backward()`` on the outputs of ``LSTM.forward``).

Round-1 status (interim): the forward of every step runs on the HIP kernels (``LSTM.step``); the
backward is an explicit reverse sweep over the steps with activation recomputation.  All contractions (dgrad /
wgrad GEMMs of the LSTM cell, the embedding MLP, Hidden2Normal, the social encoding) run on the fp32 MFMA GEMM
(``tnp_linear_forward``) with transposed operands; the pointwise derivatives and the index bookkeeping are plain
device-tensor expressions, and the grid scatter's backward uses ``tnp_pool_pair_cells`` (every in-range neighbour
receives the gradient of its cell, overwritten duplicates included -- SURVEY.md 8a quirk 4).  Positions fed back to
the decoder are detached exactly as in the reference (lstm/lstm.py:240-250), hidden states pooled from neighbours are
not (lstm/lstm.py:26), so BPTT couples the agents of a scene through the social encoding.
"""
import ctypes

import torch

from .. import _lib


def _lin(x, w, b=None, relu=False):
    return _lib.linear_forward(x, w, b, relu=relu)


def _mm(a, b_t):
    """a [M,K] @ b_t[N,K]^T on the matrix cores."""
    return _lib.linear_forward(a.contiguous(), b_t.contiguous(), None)


class SequenceFn(torch.autograd.Function):
    """rel_pred, pred = SequenceFn.apply(model, observed, goals, batch_split, truth, T_dec, extras, *params)"""

    @staticmethod
    def forward(ctx, model, observed, goals, batch_split, truth, T_dec, *params):
        dev = params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('LSTM parameters live on %s: move the model to a ROCm device (model.to("cuda")); '
                               'the MI355X path has no CPU fallback' % dev)
        observed = _lib.f32c(observed.detach(), dev)
        truth = _lib.f32c(truth.detach(), dev) if truth is not None else None
        goals_t = _lib.f32c(goals.detach(), dev) if (goals is not None and model.goal_flag) else None
        T_obs, M = observed.size(0), observed.size(1)
        idx = _lib.SceneIndex.get(batch_split, dev)
        prim = idx.starts[:-1].long()
        H = model.hidden_dim
        h = torch.zeros(M, H, device=dev)
        c = torch.zeros(M, H, device=dev)
        normals, positions, steps = [], [], []
        if T_obs == 2:
            positions = [observed[-1]]
        with torch.no_grad():
            for t in range(1, T_obs):
                o1, o2 = observed[t - 1], observed[t]
                steps.append((0, o1, o2, h, c))
                (h, c), normal = model.step(model.encoder, (h, c), o1, o2, goals_t, batch_split)
                normals.append(normal)
                positions.append(o2 + normal[:, :2])
            pt_prev = observed[-1].clone()
            prev_none = False
            for k in range(T_dec):
                if prev_none:
                    o1 = positions[-2].clone()
                else:
                    o1 = pt_prev.clone()
                    o1[prim] = positions[-2][prim]
                if truth is None:
                    o2 = positions[-1].clone()
                else:
                    o2 = truth[k].clone()
                    o2[prim] = positions[-1][prim]
                steps.append((1, o1, o2, h, c))
                (h, c), normal = model.step(model._decoder_cell(), (h, c), o1, o2, goals_t, batch_split)
                normals.append(normal)
                positions.append(o2 + normal[:, :2])
                pt_prev = o2
                prev_none = truth is None
        ctx.model, ctx.steps, ctx.idx, ctx.goals = model, steps, idx, goals_t
        ctx.batch_split = batch_split
        ctx.pos_offset = 1 if T_obs == 2 else 0
        ctx.param_names = [n for n, _ in model.named_parameters()]
        ctx.save_for_backward(*params)
        return torch.stack(normals, dim=0), torch.stack(positions, dim=0)

    @staticmethod
    def backward(ctx, d_rel, d_pred):
        model, idx = ctx.model, ctx.idx
        P = dict(zip(ctx.param_names, ctx.saved_tensors))
        grads = {n: torch.zeros_like(p) for n, p in P.items()}
        dev = d_rel.device if d_rel is not None else d_pred.device
        M, H, E = idx.M, model.hidden_dim, model.embedding_dim
        pool = model.pool
        GD = model.goal_dim if model.goal_flag else 0
        dh = torch.zeros(M, H, device=dev)
        dc = torch.zeros(M, H, device=dev)
        sizes = (idx.starts[1:] - idx.starts[:-1]).long()
        row_base = torch.repeat_interleave(idx.starts[:-1].long(), sizes).to(torch.int32)
        row_count = torch.repeat_interleave(sizes, sizes).to(torch.int32)
        L = _lib.lib()
        emb_w, emb_b = P['input_embedding.input_embeddings.0.weight'], P['input_embedding.input_embeddings.0.bias']
        wn, bn = P['hidden2normal.linear.weight'], P['hidden2normal.linear.bias']
        layers = pool.embedding_layers() if pool is not None else []
        lay_names = []
        if pool is not None:
            lay_names = ['pool.embedding.%d' % i for i, mod in enumerate(pool.embedding) if isinstance(mod, torch.nn.Linear)]

        for s in range(len(ctx.steps) - 1, -1, -1):
            dec, o1, o2, h_prev, c_prev = ctx.steps[s]
            pre = 'decoder' if dec else 'encoder'
            w_ih, w_hh = P[pre + '.weight_ih'], P[pre + '.weight_hh']
            b_ih, b_hh = P[pre + '.bias_ih'], P[pre + '.bias_hh']
            mask = ~(torch.isnan(o1[:, 0]) | torch.isnan(o2[:, 0]))
            mk = mask.unsqueeze(1).float()
            # ---------------- recompute the step's activations ----------------
            vel = torch.nan_to_num(o2 - o1) * 4.0
            emb_lin = _lin(vel, emb_w, emb_b)
            parts = [torch.relu(emb_lin), torch.zeros(M, 2, device=dev)]
            if GD:
                gd = o2 - ctx.goals
                nf = gd.norm(dim=1, keepdim=True)
                gdir = torch.nan_to_num(torch.where(nf == 0, torch.zeros_like(gd), gd / nf)) * 4.0
                g_lin = _lin(gdir, P['goal_embedding.input_embeddings.0.weight'], P['goal_embedding.input_embeddings.0.bias'])
                parts += [torch.relu(g_lin), torch.zeros(M, 2, device=dev)]
            acts = []
            enc = None
            if pool is not None:
                tid = _lib.POOL_TYPES[pool.type_]
                G, cell, half_x, half_y = pool._geometry()
                C = pool.pooling_dim
                if pool.type_ == 'social':
                    enc = _lin(h_prev, P['pool.hidden_dim_encoding.weight'], P['pool.hidden_dim_encoding.bias'])
                grid = torch.empty(M, C * G * G, device=dev)
                o1c, o2c = o1.contiguous(), o2.contiguous()
                _lib.check(L.tnp_pool_grid_forward(tid, _lib.ptr(o1c), _lib.ptr(o2c), _lib.ptr(enc), C, _lib.ptr(idx.starts),
                                                   idx.B, idx.n_max, G, C, cell, half_x, half_y, float(pool.constant),
                                                   _lib.ptr(grid), C * G * G, None, _lib.stream_ptr()), 'grid')
                x = grid
                for lin, name in zip(layers, lay_names):
                    acts.append(x)
                    x = _lin(x, P[name + '.weight'], P[name + '.bias'], relu=True)
                parts.append(x)
                pooled = x
            X = torch.cat(parts, dim=1)
            gates = _lin(X, w_ih, b_ih) + _lin(h_prev, w_hh, b_hh)
            gi, gf, gg, go = torch.sigmoid(gates[:, :H]), torch.sigmoid(gates[:, H:2 * H]), torch.tanh(gates[:, 2 * H:3 * H]), \
                torch.sigmoid(gates[:, 3 * H:])
            c_new = gf * c_prev + gi * gg
            tc = torch.tanh(c_new)
            h_out = go * tc
            # ---------------- Hidden2Normal backward (lstm/modules.py:56-64) ----------------
            dn = torch.zeros(M, 5, device=dev)
            if d_rel is not None:
                dn = dn + torch.nan_to_num(d_rel[s])
            if d_pred is not None:
                dn[:, :2] = dn[:, :2] + torch.nan_to_num(d_pred[s + ctx.pos_offset])
            dn = dn * mk
            lin_n = _lin(h_out, wn, bn)
            sg = torch.sigmoid(lin_n[:, 2:5])
            dlin = dn.clone()
            dlin[:, 2:4] = dn[:, 2:4] * 0.2 * sg[:, 0:2] * (1 - sg[:, 0:2])
            dlin[:, 4] = dn[:, 4] * 0.7 * sg[:, 2] * (1 - sg[:, 2])
            grads['hidden2normal.linear.weight'] += _mm(dlin.t(), h_out.t())
            grads['hidden2normal.linear.bias'] += dlin.sum(0)
            dh_tot = dh + _mm(dlin, wn.t())
            # ---------------- LSTMCell backward (present rows; absent rows pass the state gradient through) --------
            dh_m, dc_m = dh_tot * mk, dc * mk
            do = dh_m * tc
            dct = dc_m + dh_m * go * (1 - tc * tc)
            dG = torch.cat([dct * gg * gi * (1 - gi), dct * c_prev * gf * (1 - gf), dct * gi * (1 - gg * gg),
                            do * go * (1 - go)], dim=1)
            dG_t = dG.t().contiguous()
            grads[pre + '.weight_ih'] += _mm(dG_t, X.t())
            grads[pre + '.weight_hh'] += _mm(dG_t, h_prev.t())
            bsum = dG.sum(0)
            grads[pre + '.bias_ih'] += bsum
            grads[pre + '.bias_hh'] += bsum
            dX = _mm(dG, w_ih.t())
            dh_prev = _mm(dG, w_hh.t()) + dh_tot * (1 - mk)
            dc_prev = dct * gf + dc * (1 - mk)
            # ---------------- input / goal embedding backward ----------------
            de = dX[:, :E - 2] * (emb_lin > 0).float()
            grads['input_embedding.input_embeddings.0.weight'] += _mm(de.t(), vel.t())
            grads['input_embedding.input_embeddings.0.bias'] += de.sum(0)
            if GD:
                dg_ = dX[:, E:E + GD - 2] * (g_lin > 0).float()
                grads['goal_embedding.input_embeddings.0.weight'] += _mm(dg_.t(), gdir.t())
                grads['goal_embedding.input_embeddings.0.bias'] += dg_.sum(0)
            # ---------------- grid embedding MLP + scatter + social encoding backward ----------------
            if pool is not None:
                dy = dX[:, E + GD:]
                out_act = pooled
                for li in range(len(layers) - 1, -1, -1):
                    name = lay_names[li]
                    w = P[name + '.weight']
                    dy = dy * (out_act > 0).float()
                    grads[name + '.weight'] += _mm(dy.t(), acts[li].t())
                    grads[name + '.bias'] += dy.sum(0)
                    need_din = li > 0 or pool.type_ == 'social'
                    if need_din:
                        dy = _mm(dy, w.t())
                    out_act = acts[li]
                if pool.type_ == 'social':
                    dgrid = dy.view(M, C, G * G)
                    cells = torch.empty(M, idx.n_max, dtype=torch.int32, device=dev)
                    _lib.check(L.tnp_pool_pair_cells(_lib.ptr(o2c), _lib.ptr(row_base), _lib.ptr(row_count), M, idx.n_max, G,
                                                     cell, half_x, half_y, _lib.ptr(cells), _lib.stream_ptr()), 'pair_cells')
                    valid = cells >= 0
                    rows, js = valid.nonzero(as_tuple=True)
                    cl = cells[rows, js].long()
                    contrib = dgrid[rows, :, cl]                                  # [pairs, C]
                    denc = torch.zeros(M, C, device=dev)
                    denc.index_add_(0, row_base[rows].long() + js, contrib)
                    grads['pool.hidden_dim_encoding.weight'] += _mm(denc.t(), h_prev.t())
                    grads['pool.hidden_dim_encoding.bias'] += denc.sum(0)
                    dh_prev = dh_prev + _mm(denc, P['pool.hidden_dim_encoding.weight'].t())
            dh, dc = dh_prev, dc_prev

        # parameters the forward never touches get no gradient (None, as autograd does for the reference), so that
        # optimizers skip them: a zero gradient would still let Adam + weight decay move them
        def unused(n):
            if n.startswith('goal_embedding.') and not model.goal_flag:
                return True
            return n.startswith('pool.hidden_dim_encoding.') and (pool is None or pool.type_ != 'social')
        out = [None] * 6
        for n in ctx.param_names:
            out.append(None if unused(n) else grads[n])
        return tuple(out)


def run_sequence_with_grad(model, observed, goals, batch_split, truth, T_dec):
    params = [p for _, p in model.named_parameters()]
    return SequenceFn.apply(model, observed, goals, batch_split, truth, T_dec, *params)
