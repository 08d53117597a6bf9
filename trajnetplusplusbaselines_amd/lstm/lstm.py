"""trajnetbaselines.lstm.LSTM / LSTMPredictor on MI355X.

Same class names, constructor arguments, method signatures, return shapes and state_dict keys as the
reference's lstm/lstm.py:45-313, so reference checkpoints load and the reference trainer / evaluator call
sites keep working; the recurrent step runs in hand-written HIP (csrc/lstm_seq.hip and friends) on dense
[M,H] state instead of the reference's Python lists of per-track tensors.  There is no CPU fallback.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from .. import data as trajdata
from .modules import Hidden2Normal, InputEmbedding

NAN = float('nan')


def drop_distant(xy, r=6.0):
    """Drops pedestrians more than r meters away from primary ped (reference lstm/lstm.py:16-22; host side)."""
    distance_2 = np.sum(np.square(xy - xy[:, 0:1]), axis=2)
    mask = np.nanmin(distance_2, axis=0) < r**2
    return xy[:, mask], mask


class LSTM(torch.nn.Module):
    def __init__(self, embedding_dim=64, hidden_dim=128, pool=None, pool_to_input=True, goal_dim=None,
                 goal_flag=False):
        """Arguments as in reference lstm/lstm.py:46-63."""
        super(LSTM, self).__init__()
        self.hidden_dim = hidden_dim
        self.embedding_dim = embedding_dim
        self.pool = pool
        self.pool_to_input = pool_to_input
        if pool is not None and not pool_to_input and pool.out_dim != hidden_dim:
            raise ValueError('pool_to_input=False adds the interaction vector to the hidden state: pool.out_dim must '
                             'equal hidden_dim (reference lstm/lstm.py:150-151)')

        scale = 4.0
        self.input_embedding = InputEmbedding(2, self.embedding_dim, scale)

        self.goal_flag = goal_flag
        self.goal_dim = goal_dim or embedding_dim
        self.goal_embedding = InputEmbedding(2, self.goal_dim, scale)
        goal_rep_dim = self.goal_dim if self.goal_flag else 0

        pooling_dim = 0
        if pool is not None and self.pool_to_input:
            pooling_dim = self.pool.out_dim

        # parameter containers only (weight_ih [4H, I], weight_hh [4H, H], gate order i,f,g,o); the cell itself
        # is the fused gates GEMM + pointwise epilogue of csrc/gemm_f32_mfma.hip
        self.encoder = torch.nn.LSTMCell(self.embedding_dim + goal_rep_dim + pooling_dim, self.hidden_dim)
        if not self._ENCODER_ONLY:
            self.decoder = torch.nn.LSTMCell(self.embedding_dim + goal_rep_dim + pooling_dim, self.hidden_dim)
            self.hidden2normal = Hidden2Normal(self.hidden_dim)

        #: kernel-variant selector forwarded to the C ABI (0 = defaults), see DESIGN.md
        self.kernel_variant = 0
        #: run the first grid-embedding layer on the sparse winner table when the configuration allows it
        self.sparse_embedding = True
        self._ws = None
        from .. import ops            # registers the torch.library ops; the handle is what trajnet::lstm_sequence looks the module up by
        self._op_handle = ops.model_handle(self)
        self._grad_reduce_fn = None   # data-parallel training: parallel.GradReducer, see lstm/train_step.py
        self._cell_major = None  # (key, tensor): cell-major copy of pool.embedding[0].weight
        self._quad_major = None  # (key, tensor): its quad-major copy (register-accumulator sparse kernel)
        #: inference forwards replayed as hipGraphs (see _forward_graphed): True = every no-grad forward, False = never,
        #: None = only where a caller asks for it (LSTMPredictor does)
        self.graph_replay = None
        self._graphs = None      # {key: _GraphedForward}

    #: subclasses that only run the encoder (the S-GAN discriminator) construct neither decoder nor Hidden2Normal, so that
    #: they draw their parameters from the RNG in the reference's order (same seed => same weights)
    _ENCODER_ONLY = False

    # class-level defaults of everything __init__ sets besides parameters: whole-object pickles are the checkpoint format
    # (LSTMPredictor.save) and __setstate__ restores __dict__ verbatim, so a module pickled by an older revision must still
    # find the attributes that later revisions read
    kernel_variant = 0
    sparse_embedding = True
    graph_replay = None
    _graphs = None
    _ws = None
    _grad_reduce_fn = None
    _cell_major = None
    _quad_major = None

    # device-side caches (workspace, re-laid-out weight copies): rebuilt lazily, never pickled / deep-copied
    _CACHES = ('_ws', '_cell_major', '_quad_major', '_dummy_head', '_grad_reduce_fn', '_desc_cache', '_plist_cache', '_graphs')

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._CACHES:
            if k in state:
                state[k] = None
        return state

    def __setstate__(self, state):
        super(LSTM, self).__setstate__(state)
        from .. import ops            # an unpickled / deep-copied module is a new object: it gets its own op handle
        self._op_handle = ops.model_handle(self)

    # ---- descriptor / workspace ---------------------------------------------------------------------
    def _decoder_cell(self):
        return self.decoder

    def _normal_head(self):
        return self.hidden2normal.linear.weight, self.hidden2normal.linear.bias

    def _named_parameter_lists(self):
        """(names, parameters) in ``named_parameters()`` order.  The module-tree walk costs ~35 us and the training forward
        needs the list twice per call, in front of its first kernel: it is kept while every module is still the same
        object in the same place with the same number of parameters, and every parameter the same object."""
        c = self.__dict__.get('_plist_cache')
        if c is not None and all(d.get(k) is child for d, k, child in c['links']) \
                and all(len(d) == n for d, n in c['sizes']) and all(d.get(k) is q for d, k, q in c['where']):
            return c['names'], c['params']
        names, params = [], []
        for n, q in self.named_parameters():
            names.append(n)
            params.append(q)
        mods = list(self.modules())
        ids = set(id(q) for q in params)
        self.__dict__['_plist_cache'] = dict(
            names=names, params=params,
            links=[(parent._modules, name, child) for parent in mods for name, child in parent._modules.items()],
            sizes=[(mod._parameters, len(mod._parameters)) for mod in mods],
            where=[(mod._parameters, name, q) for mod in mods for name, q in mod._parameters.items() if id(q) in ids])
        return names, params

    def _descriptor_signature(self):
        pool = self.pool
        grid = None
        if pool is not None and hasattr(pool, 'embedding_layers'):
            grid = (pool.type_, pool.n, pool.cell_side, pool.pooling_dim, pool.out_dim, pool.pool_size, pool.blur_size,
                    float(pool.constant), getattr(pool, 'front', None))
        return (self.embedding_dim, self.hidden_dim, bool(self.goal_flag), self.goal_dim, int(self.kernel_variant),
                bool(self.pool_to_input), bool(self.sparse_embedding), id(pool), grid)

    def _descriptor(self):
        """(tnp_lstm_model, the tensors it points into, device).  Filling the struct costs ~80 us of host time in front of
        the first kernel of every call, so it is kept: a call whose configuration is unchanged and whose parameters are
        all still the same objects in the same modules at the same addresses (Adam updates in place) gets a copy of the
        stored struct.  Only
        the two re-laid-out copies of the first embedding layer follow the parameter's VALUE; they are refreshed on every
        call (_cell_major_weight / _quad_major_weight compare the parameter's version counter).  A descriptor that points
        into anything but parameters (AttentionMLPPooling.folded, the S-GAN discriminator's dummy head) or belongs to a
        non-grid interaction module is rebuilt every time."""
        c = self.__dict__.get('_desc_cache')
        if c is not None and c['sig'] == self._descriptor_signature() \
                and all(d.get(k) is child for d, k, child in c['links']) \
                and all(d.get(k) is t and t.data_ptr() == ptr for d, k, t, ptr in c['watched']):
            m = _lib.LstmModel.from_buffer_copy(c['m'])      # callers may edit their copy (training.py clears the head)
            keep = c['keep']
            if c['relaid'] is not None:
                w, quad = c['relaid']
                cm = self._cell_major_weight(w, self.pool)
                keep = keep + [cm]
                m.Wp0_cell_major = ctypes.c_void_p(cm.data_ptr())
                if quad:
                    qm = self._quad_major_weight(w, self.pool)
                    keep.append(qm)
                    m.Wp0_quad_major = ctypes.c_void_p(qm.data_ptr())
            return m, keep, c['dev']
        return self._build_descriptor()

    def _build_descriptor(self):
        dev = self.encoder.weight_ih.device
        if dev.type != 'cuda':
            raise RuntimeError('LSTM parameters live on %s: move the model to a ROCm device (model.to("cuda")); '
                               'the MI355X path has no CPU fallback' % dev)
        keep = []
        owners = {id(q): (mod._parameters, name) for mod in self.modules() for name, q in mod._parameters.items()
                  if q is not None}
        watched, cacheable, relaid = [], [True], [None]

        def P(t):
            o = owners.get(id(t))
            if o is None or t.dtype != torch.float32 or not t.is_contiguous():
                cacheable[0] = False                          # a computed tensor, or one that is converted below
            else:
                watched.append((o[0], o[1], t, t.data_ptr()))
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            keep.append(t)
            return ctypes.c_void_p(t.data_ptr())

        m = _lib.LstmModel()
        m.E, m.H = self.embedding_dim, self.hidden_dim
        m.goal_flag, m.goal_dim = int(self.goal_flag), self.goal_dim
        ie, ge = self.input_embedding.input_embeddings[0], self.goal_embedding.input_embeddings[0]
        m.We, m.be, m.Wg, m.bg = P(ie.weight), P(ie.bias), P(ge.weight), P(ge.bias)
        for pre, cell in (('enc', self.encoder), ('dec', self._decoder_cell())):
            setattr(m, pre + '_Wih', P(cell.weight_ih))
            setattr(m, pre + '_Whh', P(cell.weight_hh))
            setattr(m, pre + '_bih', P(cell.bias_ih))
            setattr(m, pre + '_bhh', P(cell.bias_hh))
        wn, bn = self._normal_head()
        m.Wn, m.bn = P(wn), P(bn)
        m.pool_type = _lib.POOL_NONE
        m.n, m.C, m.P, m.n_layers = 0, 0, 0, 0
        pool = self.pool
        from .non_gridbased_pooling import (NearestNeighborMLP, HiddenStateMLPPooling, AttentionMLPPooling,
                                            NearestNeighborLSTM, TrajectronPooling)
        if isinstance(pool, (NearestNeighborLSTM, TrajectronPooling)):
            nnl = isinstance(pool, NearestNeighborLSTM)
            lin, cell = pool.embedding[0], pool.pool_lstm
            m.pool_type, m.P, m.C = (_lib.POOL_NNLSTM if nnl else _lib.POOL_TRAJ), pool.out_dim, 4
            m.n = pool.n if nnl else 0
            m.dims[0] = pool.hidden_dim
            m.Wp[0], m.bp[0] = P(lin.weight), P(lin.bias)
            m.Wx[0], m.bx[0] = P(cell.weight_ih), P(cell.bias_ih)
            m.Wx[1], m.bx[1] = P(cell.weight_hh), P(cell.bias_hh)
            m.Wx[2], m.bx[2] = P(pool.hidden2pool.weight), P(pool.hidden2pool.bias)
        elif isinstance(pool, NearestNeighborMLP):
            lin = pool.embedding[0]
            m.pool_type, m.n, m.C, m.P = _lib.POOL_NN, pool.n, pool.input_dim, pool.out_dim
            m.Wp[0], m.bp[0] = P(lin.weight), P(lin.bias)
        elif isinstance(pool, (HiddenStateMLPPooling, AttentionMLPPooling)):
            attn = isinstance(pool, AttentionMLPPooling)
            m.pool_type, m.P = (_lib.POOL_ATTNMLP if attn else _lib.POOL_HIDDENMLP), pool.out_dim
            if attn:
                wq, bq, wu, wfin, bfin = pool.folded()
                m.constant = float(pool.fill_value)
                m.Wx[0], m.bx[0], m.Wx[1], m.Wx[2], m.bx[2] = P(wq), P(bq), P(wu), P(wfin), P(bfin)
            m.dims[0], m.dims[1], m.dims[2] = pool.mlp_dim_spatial, pool.mlp_dim_vel, pool.mlp_dim_hidden
            m.C = pool.mlp_dim_hidden
            m.Wp[0], m.bp[0] = P(pool.spatial_embedding[0].weight), P(pool.spatial_embedding[0].bias)
            if pool.mlp_dim_vel:
                m.Wp[1], m.bp[1] = P(pool.vel_embedding[0].weight), P(pool.vel_embedding[0].bias)
            if pool.mlp_dim_hidden:
                m.Wh, m.bh = P(pool.hidden_embedding[0].weight), P(pool.hidden_embedding[0].bias)
            m.Wp[2], m.bp[2] = P(pool.out_projection.weight), P(pool.out_projection.bias)
        elif pool is not None:
            if not hasattr(pool, 'embedding_layers'):
                raise NotImplementedError('interaction module %s does not run on the MI355X path (supported: '
                                          'GridBasedPooling, NearestNeighborMLP, HiddenStateMLPPooling)'
                                          % type(pool).__name__)
            # pool_size / blur_size (never set by the trainer; reference gridbased_pooling.py:297-304): the fine grid of
            # n * pool_size cells per side is blurred and summed down to n x n by csrc/pool_grid.hip grid_finish_kernel in
            # front of the embedding MLP -- inference; the backward sweep has no such kernel (run_sequence_with_grad raises)
            m.pool_size, m.blur_size = int(pool.pool_size), int(pool.blur_size)
            m.pool_type = _lib.POOL_TYPES[pool.type_]
            G, cell, half_x, half_y = pool._geometry()
            m.n, m.C, m.P = pool.n, pool.pooling_dim, pool.out_dim
            m.cell, m.half_x, m.half_y, m.constant = cell, half_x, half_y, float(pool.constant)
            layers = pool.embedding_layers()
            if not layers:
                raise NotImplementedError("embedding_arch 'None' is not supported by the fused step")
            m.n_layers = len(layers)
            m.dims[0] = pool.pooling_dim * pool.n * pool.n
            for li, lin in enumerate(layers):
                m.dims[li + 1] = lin.weight.shape[0]
                m.Wp[li] = P(lin.weight)
                m.bp[li] = P(lin.bias)
            if pool.type_ == 'social':
                m.Wh, m.bh = P(pool.hidden_dim_encoding.weight), P(pool.hidden_dim_encoding.bias)
                if self.sparse_embedding and float(pool.constant) == 0.0 and pool.pooling_dim in (4, 8, 16, 32) \
                        and layers[0].weight.shape[0] % 4 == 0:
                    # (not through P: these follow the parameter's value and are refreshed by _descriptor on every call)
                    quad = layers[0].weight.shape[0] % 64 == 0 and pool.pooling_dim % 4 == 0
                    relaid[0] = (layers[0].weight, quad)
        m.variant = int(self.kernel_variant)
        if pool is not None and not self.pool_to_input:
            m.variant |= 1 << 17   # interaction vector added to the hidden state (lstm/lstm.py:150-151)
        grid_or_none = pool is None or hasattr(pool, 'embedding_layers')
        self.__dict__['_desc_cache'] = None
        if cacheable[0] and grid_or_none:
            # (the module tree as it is now: a replaced sub-module invalidates the stored struct like a moved parameter does)
            links = [(parent._modules, name, child) for parent in self.modules() for name, child in parent._modules.items()]
            self.__dict__['_desc_cache'] = dict(sig=self._descriptor_signature(), watched=watched, links=links, m=m,
                                                keep=list(keep), dev=dev, relaid=relaid[0])
        if relaid[0] is not None:
            w, quad = relaid[0]
            m = _lib.LstmModel.from_buffer_copy(m)
            cm = self._cell_major_weight(w, pool)
            keep.append(cm)
            m.Wp0_cell_major = ctypes.c_void_p(cm.data_ptr())
            if quad:
                qm = self._quad_major_weight(w, pool)
                keep.append(qm)
                m.Wp0_quad_major = ctypes.c_void_p(qm.data_ptr())
        return m, keep, dev

    def _relaid_weights(self, weight, pool):
        """The copies of the first embedding layer the sparse kernels read (include/trajnet_hip.h, tnp_lstm_model):
        cell-major W'[c][ch][o] = W[o][ch*n*n + c] (pool_embed_sparse.hip, the sparse backward) and, when the shape allows
        it, quad-major W''[c][o/64][ch/4][o%64][ch%4] (what the register-accumulator kernel streams: a wave's C x 64 weights
        of a cell are one contiguous block).  Both from ONE native launch (tnp_pool_embed_weight_layouts), rebuilt only
        when the parameter changes (data_ptr / in-place version)."""
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), str(weight.device))
        if self._cell_major is None or self._cell_major[0] != key:
            n1, C, ncell = weight.shape[0], pool.pooling_dim, pool.n * pool.n
            w = weight.detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.float().contiguous()
            quad = n1 % 64 == 0 and C % 4 == 0
            cm = torch.empty(ncell, C, n1, dtype=torch.float32, device=w.device)
            qm = torch.empty(ncell, n1 // 64, C // 4, 64, 4, dtype=torch.float32, device=w.device) if quad else None
            _lib.check(_lib.lib().tnp_pool_embed_weight_layouts(_lib.ptr(w), w.stride(0), n1, C, ncell, _lib.ptr(cm),
                                                                _lib.ptr(qm), _lib.stream_ptr()), 'tnp_pool_embed_weight_layouts')
            # built on THIS stream: a forward pass on another stream (two batches in flight) must wait for the launch above
            self._cell_major = (key, cm, _lib.StreamMark())
            self._quad_major = (key, qm)
        self._cell_major[2].join()
        return self._cell_major[1], self._quad_major[1]

    def _cell_major_weight(self, weight, pool):
        return self._relaid_weights(weight, pool)[0]

    def _quad_major_weight(self, weight, pool):
        return self._relaid_weights(weight, pool)[1]

    def _workspace(self, m, M, B, dev):
        need = _lib.lib().tnp_lstm_workspace_bytes(ctypes.byref(m), M, B)
        if need == 0:
            _lib.check(-1, 'tnp_lstm_workspace_bytes')
        # one workspace per STREAM: forward passes of this model on different streams (two batches in flight) must not share
        # the recurrent state and scratch buffers
        if not isinstance(self._ws, dict):
            self._ws = {}
        sk = _lib.raw_stream()                                # what _lib.stream_ptr() launches on
        ws = self._ws.get(sk)
        if ws is None or ws.numel() < need or ws.device != dev:
            if len(self._ws) >= 8:                # streams come and go: keep the most recently created workspaces only
                self._ws.pop(next(iter(self._ws)))
            ws = self._ws[sk] = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws, need

    # ---- one recurrent step (reference lstm/lstm.py:91-168) -------------------------------------------
    def step(self, lstm, hidden_cell_state, obs1, obs2, goals, batch_split, pad_to=None):
        """One masked step.  ``hidden_cell_state`` may be the reference's (list of [H] tensors, list of [H]
        tensors) or a pair of dense [M,H] tensors; the same kind is returned, with ``normal`` [M,5].
        ``pad_to``: see ``forward``."""
        m, keep, dev = self._descriptor()
        was_list = isinstance(hidden_cell_state[0], (list, tuple))
        if was_list:
            h_in = torch.stack(list(hidden_cell_state[0]), dim=0)
            c_in = torch.stack(list(hidden_cell_state[1]), dim=0)
        else:
            h_in, c_in = hidden_cell_state
        h_in, c_in = _lib.f32c(h_in, dev), _lib.f32c(c_in, dev)
        obs1, obs2 = _lib.f32c(obs1, dev), _lib.f32c(obs2, dev)
        M = obs2.size(0)
        idx = _lib.SceneIndex.get(batch_split, dev, pad_to)
        if idx.M != M:
            raise ValueError('batch_split covers %d tracks, observations have %d' % (idx.M, M))
        goals_t = _lib.f32c(goals, dev) if (goals is not None and self.goal_flag) else None
        decoder = 1 if lstm is self.decoder else 0
        ws, need = self._workspace(m, M, idx.B, dev)
        h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
        normal = torch.empty(M, 5, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().tnp_lstm_step(
            ctypes.byref(m), decoder, _lib.ptr(h_in), _lib.ptr(c_in), _lib.ptr(obs1), _lib.ptr(obs2),
            _lib.ptr(goals_t), _lib.ptr(idx.starts), idx.B, M, idx.n_max, _lib.ptr(idx.slots), _lib.ptr(h_out), _lib.ptr(c_out),
            _lib.ptr(normal), _lib.ptr(ws), need, _lib.stream_ptr()), 'tnp_lstm_step')
        if was_list:
            return (list(h_out.unbind(0)), list(c_out.unbind(0))), normal
        return (h_out, c_out), normal

    # ---- whole sequence (reference lstm/lstm.py:170-264) ----------------------------------------------
    def forward(self, observed, goals, batch_split, prediction_truth=None, n_predict=None, pad_to=None, graph=None):
        """observed [T_obs,M,2], goals [M,2], batch_split [B+1] -> (rel_pred_scene [S,M,5], pred_scene [S,M,2]).

        ``graph`` (extension): replay this inference forward as a hipGraph when the same call shape comes back (see
        ``_forward_graphed``); None = what ``self.graph_replay`` says.

        ``pad_to`` (extension, keyword only in spirit) names the number of slots the reference would have padded the
        scenes to (lstm/lstm.py:29: the padded, absent slots clobber cell (0, 0) of shorter scenes' grids and enter
        AttentionMLPPooling's softmax).  None = the largest scene of this call, i.e. exactly what the reference does
        with this batch; an int = the largest scene of the WHOLE batch when this call holds a shard of it (results then
        equal the unsharded batch bit for bit while both select the same GEMM tiles, to fp32 summation order otherwise); 'scene' = every scene unpadded, i.e. what one reference call per scene
        gives (the evaluator); or per-scene slot counts.  See ``_lib.SceneIndex``."""
        assert ((prediction_truth is None) + (n_predict is None)) == 1
        if prediction_truth is not None and isinstance(prediction_truth, (list, tuple)):
            prediction_truth = torch.stack(list(prediction_truth), dim=0)
        T_dec = prediction_truth.size(0) if prediction_truth is not None else n_predict - 1
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: same kernels step by step + an explicit backward sweep (lstm/training.py)
            trainable_pool = hasattr(self.pool, 'embedding_layers') or \
                type(self.pool).__name__ in ('NearestNeighborMLP', 'HiddenStateMLPPooling', 'AttentionMLPPooling',
                                             'NearestNeighborLSTM', 'TrajectronPooling')
            if self.pool is not None and not trainable_pool:
                raise NotImplementedError('training (backward) through %s is not available on the MI355X path yet; '
                                          'use model.eval() / torch.no_grad() for inference' % type(self.pool).__name__)
            if torch.compiler.is_compiling() and (pad_to is None or isinstance(pad_to, int)) \
                    and getattr(self, '_grad_reduce_fn', None) is None:
                # under torch.compile the training sequence is ONE dispatcher op with its backward behind a handle
                # (ops.py: trajnet::lstm_sequence_train / lstm_sequence_backward) -- no graph break
                # only the parameters the sequence touches enter the graph: the others keep .grad = None, as in eager mode
                from .training import unused_parameter_names
                skip = set(unused_parameter_names(self, T_dec))
                rel_pred, pred, _, _ = torch.ops.trajnet.lstm_sequence_train(
                    observed, goals, torch.as_tensor(batch_split), prediction_truth, T_dec, int(pad_to or 0), self._op_handle,
                    [p for n, p in self.named_parameters() if n not in skip])
                return rel_pred, pred
            from .training import run_sequence_with_grad
            opts = {'pad_to': pad_to, 'reduce_fn': getattr(self, '_grad_reduce_fn', None)}
            rel_pred, pred, _ = run_sequence_with_grad(self, observed, goals, batch_split, prediction_truth, T_dec, opts)
            return rel_pred, pred
        if torch.compiler.is_compiling() and (pad_to is None or isinstance(pad_to, int)):
            # under torch.compile the whole sequence is ONE dispatcher op (ops.py: trajnet::lstm_sequence) -- no graph break
            return torch.ops.trajnet.lstm_sequence(observed, goals, torch.as_tensor(batch_split), prediction_truth, T_dec,
                                                   int(pad_to or 0), self._op_handle, list(self.parameters()))
        use_graph = self.graph_replay if self.graph_replay is not None else bool(graph)
        if use_graph and not torch.is_grad_enabled() and _GRAPHS_ALLOWED:
            return self._forward_graphed(observed, goals, batch_split, prediction_truth, T_dec, pad_to)
        rel_pred, pred, _ = self._run_sequence(observed, goals, batch_split, prediction_truth, T_dec, pad_to=pad_to)
        return rel_pred, pred

    # ---- inference forward as a hipGraph ----------------------------------------------------------------
    _GRAPH_AFTER = 2          # calls of one shape that run eagerly before it is captured
    _GRAPH_MAX = 24           # captured shapes kept (least recently used goes first) ...
    _GRAPH_MAX_BYTES = 4 << 30   # ... and the bytes of their private workspaces / static buffers

    def _forward_graphed(self, observed, goals, batch_split, truth, T_dec, pad_to):
        """One inference forward = 4 launches x (T_obs - 1 + T_dec) steps enqueued by ``tnp_lstm_forward``: ~80 launches, ~0.35 ms of
        host time (0.25 ms inside the native call).  The sequence driver never synchronises or allocates, so it can be captured once per call shape -- (model
        weights' addresses and configuration, scene structure, sequence lengths, decoder mode, stream) -- into a hipGraph over
        static input / output buffers and a private workspace; later calls of that shape copy their inputs in, replay the graph
        (0.11 ms of host time) and clone the outputs.  A shape is captured on its third call (one-off shapes stay eager); the
        kernels, their order and their arguments are those of the eager path, so the outputs are bit-identical
        (tests/test_gpu_graph.py).

        OPT-IN (``graph=True`` / ``self.graph_replay``), because on MI355X with ROCm 7.2 the replayed kernel nodes run slower on
        the device than the same launches enqueued one by one (tools/diag/small_batch_latency.py: 64 x 32 agents 1.45 ms per
        forward replayed against 1.31 ms; one 4-agent scene 0.88 against 0.76 ms): what the replay buys is host time, which pays
        when several SMALL batches are in flight on their own streams (8 x 32 agents: 0.72 ms per forward with two graphs in
        flight against 1.26 ms eager; 1 x 32: 0.64 against 1.04 ms) or when the host has other work to do."""
        m, keep, dev = self._descriptor()
        idx = _lib.SceneIndex.get(batch_split, dev, pad_to)
        T_obs, M = observed.size(0), observed.size(1)
        if idx.M != M:
            raise ValueError('batch_split covers %d tracks, observed has %d' % (idx.M, M))
        goals_t = goals if (goals is not None and self.goal_flag) else None
        # (the caller's stream is part of the key: forwards on different streams -- batches in flight -- must not share buffers)
        # ... and the scene structure enters by CONTENT (SceneIndex.key), not by object identity: clearing SceneIndex._cache
        # must not orphan captured graphs (they would pin their private pools until evicted)
        key = (bytes(m), getattr(idx, 'key', idx), T_obs, T_dec, truth is not None, goals_t is not None,
               _lib.raw_stream())
        if not isinstance(self._graphs, dict):
            self._graphs = {}
        e = self._graphs.get(key)
        if not isinstance(e, _GraphedForward):
            seen = int(e or 0) + 1
            if seen <= self._GRAPH_AFTER or M == 0:
                self._graphs[key] = seen
                if len(self._graphs) > 8 * self._GRAPH_MAX:       # shapes that never came back
                    for k in [k for k, v in self._graphs.items() if not isinstance(v, _GraphedForward)][:4 * self._GRAPH_MAX]:
                        del self._graphs[k]
                rel_pred, pred, _ = self._run_sequence(observed, goals_t, batch_split, truth, T_dec, pad_to=pad_to)
                return rel_pred, pred
            e = _GraphedForward(self, keep, _lib.f32c(observed, dev), _lib.f32c(goals_t, dev) if goals_t is not None else None,
                                batch_split, _lib.f32c(truth, dev) if truth is not None else None, T_dec, pad_to)
            self._graphs.pop(key, None)
            self._graphs[key] = e
            live = [(k, v) for k, v in self._graphs.items() if isinstance(v, _GraphedForward)]
            while len(live) > self._GRAPH_MAX or (len(live) > 1 and sum(v.nbytes for _, v in live) > self._GRAPH_MAX_BYTES):
                del self._graphs[live.pop(0)[0]]
        else:
            self._graphs[key] = self._graphs.pop(key)             # most recently used last
        return e(observed, goals_t, truth)

    def forward_with_loss(self, observed, goals, batch_split, targets, criterion, prediction_truth=None, n_predict=None,
                          pad_to=None):
        """``forward`` + ``criterion(rel_pred[-T:], targets, batch_split)`` in one pass (no gradients): the kernel that
        finishes a step evaluates the primaries' loss while the step's normal is still in registers (SURVEY.md 8f rank 3:
        what Trainer.val_batch computes, lstm/trainer.py:296-309).  ``criterion`` is a PredictionLoss or L2Loss (their
        auxiliary collision term needs predicted positions and is not part of the validation loss in the reference
        either).  Returns (rel_pred, pred, loss) with loss exactly what the criterion returns."""
        from .loss import PredictionLoss, L2Loss
        assert ((prediction_truth is None) + (n_predict is None)) == 1
        if isinstance(criterion, PredictionLoss):
            mode, bg, scale = 0, criterion.background_rate, float(criterion.loss_multiplier)
        elif isinstance(criterion, L2Loss):
            mode, bg, scale = 1, 0.0, 0.5 * criterion.loss_multiplier
        else:
            raise TypeError('forward_with_loss takes a PredictionLoss or an L2Loss')
        if prediction_truth is not None and isinstance(prediction_truth, (list, tuple)):
            prediction_truth = torch.stack(list(prediction_truth), dim=0)
        T_dec = prediction_truth.size(0) if prediction_truth is not None else n_predict - 1
        rel_pred, pred, rows = self._run_sequence(observed, goals, batch_split, prediction_truth, T_dec, pad_to=pad_to,
                                                  fused_loss=(targets, mode, bg))
        dev = rel_pred.device
        idx = _lib.SceneIndex.get(batch_split, dev)
        keep = bool(criterion.keep_batch_dim)
        T = rows.size(0)
        out = torch.empty(idx.B if keep else 1, dtype=torch.float32, device=dev)
        ws = torch.empty(T * idx.B, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().tnp_primary_loss_reduce(_lib.ptr(rows), rows.size(1), _lib.ptr(idx.starts), idx.B, T, int(keep),
                                                      scale, _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()),
                   'tnp_primary_loss_reduce')
        return rel_pred, pred, (out if keep else out[0])

    def _run_sequence(self, observed, goals, batch_split, truth, T_dec, w_ctx=None, b_ctx=None, noise=None,
                      want_h_final=False, pad_to=None, fused_loss=None, h_scale=None, private_ws=None):
        """tnp_lstm_forward(_ex): T_obs-1 encoder steps + T_dec decoder steps; optional S-GAN hooks.
        ``fused_loss`` = (targets [T_loss, M, 2], mode, background_rate): the per-primary loss values of the last T_loss
        outputs are evaluated inside the sequence (returned in place of h_final as [T_loss, M] rows)."""
        m, keep, dev = self._descriptor()
        observed = _lib.f32c(observed, dev)
        T_obs, M = observed.size(0), observed.size(1)
        idx = _lib.SceneIndex.get(batch_split, dev, pad_to)
        if idx.M != M:
            raise ValueError('batch_split covers %d tracks, observed has %d' % (idx.M, M))
        truth = _lib.f32c(truth, dev) if truth is not None else None
        goals_t = _lib.f32c(goals, dev) if (goals is not None and self.goal_flag) else None
        n_steps = T_obs - 1 + T_dec
        npos = n_steps + (1 if T_obs == 2 else 0)
        rel_pred = torch.empty(n_steps, M, 5, dtype=torch.float32, device=dev)
        pred = torch.empty(npos, M, 2, dtype=torch.float32, device=dev)
        if private_ws is not None:       # hipGraph capture: the workspace belongs to the graph (private_ws: a list that receives it)
            need = _lib.lib().tnp_lstm_workspace_bytes(ctypes.byref(m), M, idx.B)
            if need == 0:
                _lib.check(-1, 'tnp_lstm_workspace_bytes')
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            private_ws.append(ws)
        else:
            ws, need = self._workspace(m, M, idx.B, dev)
        ex = _lib.LstmExtras()
        h_final = None
        if noise is not None:
            w_ctx, b_ctx, noise = _lib.f32c(w_ctx.detach(), dev), _lib.f32c(b_ctx.detach(), dev), _lib.f32c(noise, dev)
            ex.W_ctx, ex.b_ctx, ex.noise = _lib.ptr(w_ctx), _lib.ptr(b_ctx), _lib.ptr(noise)
            ex.noise_dim = int(noise.shape[-1])
            groups = noise.numel() // ex.noise_dim       # [k, noise_dim]: k samples batched as k replicas of the scenes
            if groups > 1:
                if M % groups:
                    raise ValueError('%d noise vectors for %d tracks' % (groups, M))
                ex.noise_group_tracks = M // groups
        if h_scale is not None:      # VAE: hidden <- hidden * h_scale [M, H] after the last encoder step
            h_scale = _lib.f32c(h_scale, dev)
            if tuple(h_scale.shape) != (M, self.hidden_dim):
                raise ValueError('h_scale must be [%d, %d]' % (M, self.hidden_dim))
            ex.h_scale = _lib.ptr(h_scale)
        if want_h_final:
            h_final = torch.empty(M, self.hidden_dim, dtype=torch.float32, device=dev)
            ex.h_final = _lib.ptr(h_final)
        loss_rows = None
        if fused_loss is not None:
            tgt, mode, bg = fused_loss
            tgt = _lib.f32c(tgt, dev)
            if tgt.dim() != 3 or tgt.size(1) != M or tgt.size(2) != 2:
                raise ValueError('fused loss targets must be [T_loss, M, 2]')
            loss_rows = torch.empty(tgt.size(0), M, dtype=torch.float32, device=dev)
            ex.loss_targets, ex.loss_values = _lib.ptr(tgt), _lib.ptr(loss_rows)
            ex.loss_steps, ex.loss_mode, ex.loss_background_rate = int(tgt.size(0)), int(mode), float(bg)
        _lib.check(_lib.lib().tnp_lstm_forward_ex(
            ctypes.byref(m), _lib.ptr(observed), T_obs, M, _lib.ptr(goals_t), _lib.ptr(idx.starts),
            _lib.ptr(idx.primary), idx.B, idx.n_max, _lib.ptr(idx.slots), _lib.ptr(truth), T_dec, _lib.ptr(rel_pred), _lib.ptr(pred),
            _lib.ptr(ws), need, ctypes.byref(ex), _lib.stream_ptr()), 'tnp_lstm_forward')
        if fused_loss is not None:
            return rel_pred, pred, loss_rows
        return rel_pred, pred, h_final


#: kill switch for the hipGraph replay of inference forwards (TNP_NO_GRAPHS=1)
_GRAPHS_ALLOWED = os.environ.get('TNP_NO_GRAPHS', '0') in ('', '0')


class _GraphedForward(object):
    """One captured inference forward of one call shape: static inputs, the outputs and the workspace the captured launches
    point into, and the instantiated hipGraph (LSTM._forward_graphed).

    The graph is captured AND replayed on a stream of its own; the caller's stream is joined by events on both sides (a
    replay then never depends on which stream -- the default one included -- the caller happens to be on).  The captured
    sequence holds kernel nodes only: with ROCm 7.2 / torch 2.10 a hipMemsetAsync captured as a memset node clears its whole
    range on the first replay and less than half of it afterwards (tools/diag/graph_memset_probe.py; first seen as a vanilla
    LSTM whose third replay returned outputs that belonged to no input, tools/diag/graph_race_probe.py), which is why
    csrc/lstm_seq.hip fills and copies with kernels of its own."""

    def __init__(self, model, keep, observed, goals, batch_split, truth, T_dec, pad_to):
        self.keep = keep                     # the weight tensors the captured kernel arguments point into
        # ... and the scene-index tables (starts / primary / slots): the graph is keyed by their CONTENT, so nothing else keeps
        # this object alive once SceneIndex._cache is cleared -- the captured kernels would then read freed memory
        self.idx = _lib.SceneIndex.get(batch_split, observed.device, pad_to)
        self.obs = observed.clone()
        self.goals = goals.clone() if goals is not None else None
        self.truth = truth.clone() if truth is not None else None
        self.stream = torch.cuda.Stream(device=observed.device)
        ws = []
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):   # (synchronises the device first; private memory pool)
            self.rel, self.pred, _ = model._run_sequence(self.obs, self.goals, batch_split, self.truth, T_dec, pad_to=pad_to,
                                                         private_ws=ws)
        self.ws = ws[0]
        self.nbytes = sum(t.numel() * t.element_size() for t in (self.obs, self.goals, self.truth, self.rel, self.pred, self.ws)
                          if t is not None)
        self.replays = 0

    def __call__(self, observed, goals, truth):
        cur = torch.cuda.current_stream(self.obs.device)
        self.stream.wait_stream(cur)                             # the caller's inputs, and the previous call's clones
        with torch.cuda.stream(self.stream):
            self.obs.copy_(observed, non_blocking=True)
            if self.goals is not None:
                self.goals.copy_(goals, non_blocking=True)
            if self.truth is not None:
                self.truth.copy_(truth, non_blocking=True)
            self.graph.replay()
            rel, pred = self.rel.clone(), self.pred.clone()
        cur.wait_stream(self.stream)
        rel.record_stream(cur)
        pred.record_stream(cur)
        self.replays += 1
        return rel, pred


def _reference_pickle_module():
    """A ``pickle``-like module for ``torch.load`` whose Unpickler maps the reference's class paths
    (``trajnetbaselines.lstm.lstm.LSTM``, ``...gridbased_pooling.GridBasedPooling``, ``...sgan.sgan.SGAN``, ...) to this
    package's mirrors; everything else resolves as usual."""
    import pickle
    import types

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == 'trajnetbaselines' or module.startswith('trajnetbaselines.'):
                module = __name__.split('.')[0] + module[len('trajnetbaselines'):]
            return super(Unpickler, self).find_class(module, name)

    mod = types.ModuleType('trajnet_reference_pickle')
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.loads = lambda b, **kw: Unpickler(__import__('io').BytesIO(b), **kw).load()
    for k in ('dump', 'dumps', 'Pickler', 'PickleError', 'PicklingError', 'UnpicklingError', 'HIGHEST_PROTOCOL', 'DEFAULT_PROTOCOL'):
        setattr(mod, k, getattr(pickle, k))
    return mod


class LSTMPredictor(object):
    """Reference lstm/lstm.py:266-313: pickle-compatible wrapper used by the evaluator."""

    #: replay repeated call shapes as hipGraphs (LSTM._forward_graphed).  OFF by default: measured on MI355X / ROCm 7.2 a replay
    #: frees the host (0.11 ms instead of ~0.35 ms per forward) but the graph's kernel nodes run SLOWER on the
    #: device than the same launches enqueued one by one (+1.7 us per node at 64 x 32, ~11 us per node for the tiny kernels of a
    #: 4-agent scene: 0.91 ms per call against 0.82 ms), so it pays only with several small batches in flight (docs/history.md section 9)
    graph_replay = False

    def __init__(self, model):
        self.model = model

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_streams', None)          # predict_batches' HIP streams are not part of the pickle
        return state

    def _graph_kw(self):
        # only the plain LSTM.forward takes ``graph``: LSTMGenerator / LSTMDiscriminator / the VAE's cells subclass LSTM but
        # override forward() without it
        return {'graph': True} if (self.graph_replay and type(self.model).forward is LSTM.forward) else {}

    def _deterministic(self):
        """the plain LSTM in eval mode has no source of randomness: its `modes` forwards of one scene are the same forward"""
        return type(self.model) is LSTM

    def save(self, state, filename):
        with open(filename, 'wb') as f:
            torch.save(self, f)
        with open(filename + '.state', 'wb') as f:
            torch.save(state, f)

    @staticmethod
    def load(filename):
        """Reference lstm/lstm.py:279-282.  A whole-object pickle written by the REFERENCE (``trajnetbaselines.lstm.lstm.
        LSTMPredictor`` holding ``trajnetbaselines.lstm.*`` modules) loads too: class paths under ``trajnetbaselines.`` resolve to
        the mirrors of this package (same attribute names, same state_dict keys), so existing ``.pkl`` checkpoints are a
        drop-in (tests/test_predictor_pickle.py)."""
        with open(filename, 'rb') as f:
            return torch.load(f, weights_only=False, pickle_module=_reference_pickle_module())

    def __call__(self, paths, scene_goal, n_predict=12, modes=1, predict_all=True, obs_length=9, start_length=0,
                 args=None):
        if self.model.training:      # (eval() walks every sub-module: 0.08 ms of a 0.65 ms per-scene call)
            self.model.eval()
        with torch.no_grad():
            xy = trajdata.paths_to_xy(paths)
            batch_split = [0, xy.shape[1]]
            normalize = bool(getattr(args, 'normalize_scene', False))
            if normalize:
                xy, rotation, center, scene_goal = trajdata.center_scene(xy, obs_length, goals=scene_goal)
            xy = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float32))
            scene_goal = torch.from_numpy(np.ascontiguousarray(scene_goal, dtype=np.float32))
            batch_split = torch.tensor(batch_split, dtype=torch.int64)

            multimodal_outputs = {}
            first = None
            for num_p in range(modes):
                if first is not None and self._deterministic():
                    output_scenes = first.copy()      # (the reference runs the same deterministic forward `modes` times, lstm/lstm.py:299-311)
                else:
                    _, output_scenes = self.model(xy[start_length:obs_length], scene_goal, batch_split,
                                                  n_predict=n_predict, **self._graph_kw())
                    output_scenes = first = output_scenes.cpu().numpy()
                if normalize:
                    output_scenes = trajdata.inverse_scene(output_scenes, rotation, center)
                output_primary = output_scenes[-n_predict:, 0]
                output_neighs = output_scenes[-n_predict:, 1:]
                multimodal_outputs[num_p] = [output_primary, output_neighs]
        return multimodal_outputs

    def predict_batch(self, scenes, n_predict=12, modes=1, obs_length=9, start_length=0, args=None):
        """Batched form of ``__call__`` for the evaluator: ``scenes`` is a list of
        ``(paths, scene_goal)``; all scenes go through ONE ``LSTM.forward`` (tracks concatenated, ``batch_split``
        marking the primaries) instead of one call per scene per joblib worker (reference
        evaluator/trajnet_evaluator.py:61-75).  Returns a list (scene order) of the ``multimodal_outputs`` dicts
        ``__call__`` returns.  Scene preprocessing (paths_to_xy, center_scene) is per scene on the host, as in
        the reference; scenes never interact (lstm/lstm.py:243-250) and the forward runs with ``pad_to='scene'`` (no padded
        slots: a per-scene call has none, whereas the reference's ragged batch would clobber cell (0, 0) of the shorter
        scenes' grids), so the result equals the per-scene calls."""
        xys, goals = [], []
        for paths, scene_goal in scenes:
            xy = trajdata.paths_to_xy(paths)
            xys.append(xy)
            goals.append(np.zeros((xy.shape[1], 2)) if scene_goal is None else np.asarray(scene_goal))
        if not xys:
            return []
        handle = self.predict_xy_launch(xys, goals, n_predict, modes, obs_length, start_length, args)
        return self._unpack(handle[0], handle[1], handle[2], handle[3], n_predict)

    def predict_xy_launch(self, xys, goals, n_predict=12, modes=1, obs_length=9, start_length=0, args=None):
        """The array-level half of ``predict_batch``: ``xys`` = per-scene float64 [T, N_s, 2] arrays (what ``paths_to_xy`` gives,
        primary in column 0), ``goals`` = per-scene [N_s, 2].  Queues the ``modes`` forward passes of the batch and returns a
        handle WITHOUT waiting for the GPU; ``predict_xy_finish(handle, n_predict)`` reads the predictions back.
        ``data.predict_dataset`` builds the arrays straight from the columns of the test file and keeps a batch in flight while
        it prepares the next and writes the previous one."""
        if self.model.training:      # (eval() walks every sub-module: 0.08 ms of a 0.65 ms per-scene call)
            self.model.eval()
        normalize = bool(getattr(args, 'normalize_scene', False))
        obs_list, goal_list, frames = [], [], []
        for xy, scene_goal in zip(xys, goals):
            if xy.shape[0] < obs_length:
                raise ValueError('scene has %d frames, need at least obs_length=%d' % (xy.shape[0], obs_length))
            if normalize:
                xy, rotation, center, scene_goal = trajdata.center_scene(xy, obs_length, goals=scene_goal)
                frames.append((rotation, center))
            obs_list.append(np.asarray(xy)[start_length:obs_length])
            goal_list.append(scene_goal)
        xy, split = trajdata.batch_scenes(obs_list)
        with torch.no_grad():
            obs = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float32))   # (numpy converts: torch.tensor() of a float64 array wakes the CPU thread pool, ~5 ms on a 256-core host)
            goal = torch.from_numpy(np.ascontiguousarray(np.concatenate(goal_list, axis=0), dtype=np.float32))
            batch_split = torch.from_numpy(np.asarray(split, dtype=np.int64))
            outputs = [self.model(obs, goal, batch_split, n_predict=n_predict, pad_to='scene', **self._graph_kw())[1]
                       for _ in range(1 if self._deterministic() else modes)]
            outputs = outputs * modes if len(outputs) < modes else outputs      # deterministic model: one forward serves every mode
        return outputs, split, frames, normalize

    def predict_xy_finish(self, handle, n_predict=12):
        """-> (float64 [modes, n_predict, M, 2] predictions in world coordinates, split): the values ``predict_batch`` returns
        per scene, as one array (column ``split[s]`` = the primary of scene ``s``)."""
        outputs, split, frames, normalize = handle
        out = np.stack([o[-n_predict:].cpu().numpy() for o in outputs]).astype(np.float64)
        if normalize:
            for s in range(len(split) - 1):
                for m in range(out.shape[0]):
                    out[m, :, split[s]:split[s + 1]] = trajdata.inverse_scene(out[m, :, split[s]:split[s + 1]], *frames[s])
        return out, split

    @staticmethod
    def _unpack(outputs, split, frames, normalize, n_predict):
        """device outputs of the `modes` forward passes of one batch -> the per-scene multimodal_outputs dicts"""
        results = [dict() for _ in range(len(split) - 1)]
        for num_p, output in enumerate(outputs):
            output = output.cpu().numpy()
            for s in range(len(split) - 1):
                out = output[:, split[s]:split[s + 1]]
                if normalize:
                    out = trajdata.inverse_scene(out, *frames[s])
                results[s][num_p] = [out[-n_predict:, 0], out[-n_predict:, 1:]]
        return results

    def predict_batches(self, batches, n_predict=12, modes=1, obs_length=9, start_length=0, args=None, in_flight=2):
        """``predict_batch`` over MANY batches (an evaluation set cut into chunks of scenes) with ``in_flight`` of them on the
        GPU at a time, each on its own HIP stream: the kernels of one recurrent step run in lockstep with a fixed prologue /
        epilogue each, and the kernels of another, independent forward pass fill those gaps (two 64 x 32 Social-LSTM batches in
        flight: 1.16 ms per forward against 1.31 ms one after the other, tools/diag/two_stream_probe.py).  Results are those
        of ``predict_batch`` on every batch, bit for bit: per-stream workspaces, stream-aware caches (_lib.StreamMark).
        Returns a list (batch order) of ``predict_batch`` results."""
        if self.model.training:      # (eval() walks every sub-module: 0.08 ms of a 0.65 ms per-scene call)
            self.model.eval()
        normalize = bool(getattr(args, 'normalize_scene', False))
        dev = next(self.model.parameters()).device
        n_streams = max(1, int(in_flight))
        pool = self.__dict__.setdefault('_streams', [])        # reused across calls (each stream owns a workspace of the model)
        while len(pool) < n_streams:
            pool.append(torch.cuda.Stream(device=dev))
        streams = pool[:n_streams]
        start = torch.cuda.Event()
        start.record(torch.cuda.current_stream(dev))
        pending = []
        for bi, scenes in enumerate(batches):
            xys, goals, frames = [], [], []
            for paths, scene_goal in scenes:
                xy = trajdata.paths_to_xy(paths)
                if xy.shape[0] < obs_length:
                    raise ValueError('scene has %d frames, need at least obs_length=%d' % (xy.shape[0], obs_length))
                scene_goal = np.zeros((xy.shape[1], 2)) if scene_goal is None else np.asarray(scene_goal)
                if normalize:
                    xy, rotation, center, scene_goal = trajdata.center_scene(xy, obs_length, goals=scene_goal)
                    frames.append((rotation, center))
                xys.append(np.asarray(xy)[start_length:obs_length])
                goals.append(scene_goal)
            if not xys:
                pending.append(None)
                continue
            xy, split = trajdata.batch_scenes(xys)
            st = streams[bi % len(streams)]
            st.wait_event(start)                                   # whatever the caller had queued before this call
            with torch.no_grad(), torch.cuda.stream(st):
                obs = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float32))   # (numpy converts: torch.tensor() of a float64 array wakes the CPU thread pool, ~5 ms on a 256-core host)
                goal = torch.from_numpy(np.ascontiguousarray(np.concatenate(goals, axis=0), dtype=np.float32))
                batch_split = torch.tensor(split, dtype=torch.int64)
                outputs = [self.model(obs, goal, batch_split, n_predict=n_predict, pad_to='scene', **self._graph_kw())[1]
                           for _ in range(1 if self._deterministic() else modes)]
                outputs = outputs * modes if len(outputs) < modes else outputs
            pending.append((outputs, split, frames))
        for st in streams:
            st.synchronize()
        return [[] if item is None else self._unpack(item[0], item[1], item[2], normalize, n_predict) for item in pending]
