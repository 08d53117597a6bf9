"""The reference's VAE forecaster (vae/vae.py:26-398) on the MI355X path.

``VAE.step`` is a verbatim copy of ``LSTM.step`` in the reference (vae/vae.py:109-186 == lstm/lstm.py:91-168), so the three
recurrent passes of ``VAE.forward`` run on the fused HIP sequence driver of ``lstm/lstm.py``:

  * the observation encoder (``obs_encoder``) and the decoder (``decoder``) are one sequence with a hook between them:
    after the last encoder step the hidden state is multiplied with ``vae_decoder(z)`` (add_noise, vae/vae.py:89-107;
    ``tnp_lstm_extras.h_scale``), the cell state is kept;
  * the ``num_modes`` decoder passes run as replicated scenes of that one sequence (scenes never interact), replica k
    carrying its own multiplier;
  * in training, the prediction encoder (``pred_encoder``) is an encoder-only sequence over [observed[-1], truth ...] and
    the observation encoder runs once more on its own, so that the latent statistics ``vae_encoder_xy([h_obs | h_pred])``
    exist before the hook needs ``vae_decoder(z)``; autograd sums the two uses of the shared encoder weights.

Constructor arguments, attribute names, state_dict keys and return values are the reference's; random numbers are drawn
where and how the reference draws them (training: one ``torch`` normal tensor per mode on the host; evaluation:
``numpy.random.multivariate_normal`` per track), so a seeded run reproduces the reference's samples."""
import numpy as np
import torch

from .. import _lib
from .. import data as trajdata
from ..lstm.lstm import LSTM, drop_distant  # noqa: F401
from ..lstm.modules import Hidden2Normal, InputEmbedding


def sample_multivariate_distribution(mean, var_log):
    """reference vae/utils.py:4-23: one numpy.random.multivariate_normal draw per track, diagonal covariance exp(var_log)."""
    mean, var_log = mean.detach().cpu(), var_log.detach().cpu()
    samples = torch.zeros_like(mean)
    for track in range(mean.size(0)):
        cov_matrix = np.diag(torch.exp(var_log[track, :]).numpy())
        samples[track, :] = torch.Tensor(np.random.multivariate_normal(mean[track, :].numpy(), cov_matrix))
    return samples


class VAEEncoder(torch.nn.Module):
    """reference vae/vae.py:317-331"""

    def __init__(self, input_dim, output_dim):
        super(VAEEncoder, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.fc_mu = torch.nn.Linear(in_features=self.input_dim, out_features=self.output_dim // 2)
        self.fc_var = torch.nn.Linear(in_features=self.input_dim, out_features=self.output_dim // 2)
        self.relu = torch.nn.ReLU()

    def forward(self, inputs):
        if isinstance(inputs, (list, tuple)):
            inputs = torch.stack(list(inputs))
        inputs = torch.reshape(inputs, shape=(-1, self.input_dim))
        z_mu = torch.ops.trajnet.linear(inputs, self.fc_mu.weight, self.fc_mu.bias, True)
        z_log_var = 0.01 + torch.ops.trajnet.linear(inputs, self.fc_var.weight, self.fc_var.bias, True)
        return z_mu, z_log_var


class VAEDecoder(torch.nn.Module):
    """reference vae/vae.py:333-344"""

    def __init__(self, input_dim, output_dim):
        super(VAEDecoder, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.fc = torch.nn.Linear(in_features=self.input_dim, out_features=self.output_dim)
        self.relu = torch.nn.ReLU()

    def forward(self, inputs):
        inputs = torch.reshape(inputs, shape=(-1, self.input_dim))
        return torch.ops.trajnet.linear(inputs, self.fc.weight, self.fc.bias, True)


class _Cells(LSTM):
    """An LSTM sequence runner whose sub-modules ARE the VAE's (shared objects, nothing copied): `encoder` / `decoder` are
    two of the VAE's three LSTMCells.  Not registered as a sub-module of the VAE (its state_dict stays the reference's)."""

    def __init__(self, vae, encoder, decoder):
        with torch.random.fork_rng(devices=[]):       # LSTM.__init__ draws default weights: keep the caller's RNG stream
            super(_Cells, self).__init__(embedding_dim=vae.embedding_dim, hidden_dim=vae.hidden_dim, pool=vae.pool,
                                         pool_to_input=vae.pool_to_input, goal_dim=vae.goal_dim, goal_flag=vae.goal_flag)
        self.input_embedding, self.goal_embedding = vae.input_embedding, vae.goal_embedding
        self.encoder, self.decoder, self.hidden2normal = encoder, decoder, vae.hidden2normal


class VAE(torch.nn.Module):
    def __init__(self, embedding_dim=64, hidden_dim=128, pool=None, pool_to_input=True, goal_dim=None, goal_flag=False,
                 num_modes=1, latent_dim=128):
        """Arguments as in reference vae/vae.py:27-87 (sub-modules are created in the reference's order: a seeded default
        init gives the reference's weights)."""
        super(VAE, self).__init__()
        self.hidden_dim = hidden_dim
        self.embedding_dim = embedding_dim
        self.pool = pool
        self.pool_to_input = pool_to_input
        scale = 4.0
        self.input_embedding = InputEmbedding(2, self.embedding_dim, scale)
        self.goal_flag = goal_flag
        self.goal_dim = goal_dim or embedding_dim
        self.goal_embedding = InputEmbedding(2, self.goal_dim, scale)
        goal_rep_dim = self.goal_dim if self.goal_flag else 0
        pooling_dim = 0
        if pool is not None and self.pool_to_input:
            pooling_dim = self.pool.out_dim
        in_dim = self.embedding_dim + goal_rep_dim + pooling_dim
        self.obs_encoder = torch.nn.LSTMCell(in_dim, self.hidden_dim)
        self.pred_encoder = torch.nn.LSTMCell(in_dim, self.hidden_dim)
        self.decoder = torch.nn.LSTMCell(in_dim, self.hidden_dim)
        self.hidden2normal = Hidden2Normal(self.hidden_dim)
        self.latent_dim = latent_dim
        self.num_modes = num_modes
        self.desire = True
        self.vae_encoder_xy = VAEEncoder(2 * self.hidden_dim, 2 * self.latent_dim)
        self.vae_encoder_x = VAEEncoder(self.hidden_dim, 2 * self.latent_dim)
        self.vae_decoder = VAEDecoder(self.latent_dim, self.hidden_dim)

    # ---- the two sequence runners over the VAE's own cells (built lazily, never part of state_dict / pickles) ----
    def _runners(self):
        r = self.__dict__.get('_cells')
        if r is None or r[0].encoder is not self.obs_encoder or r[1].encoder is not self.pred_encoder \
                or r[0].decoder is not self.decoder or r[0].pool is not self.pool:
            r = (_Cells(self, self.obs_encoder, self.decoder), _Cells(self, self.pred_encoder, self.decoder))
            self.__dict__['_cells'] = r
        for c in r:
            c.training = self.training
        return r

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_cells', None)
        return state

    @staticmethod
    def _replicate(batch_split, k):
        split = torch.as_tensor(batch_split, dtype=torch.int64).cpu()
        M = int(split[-1])
        return torch.cat([split[:1]] + [split[1:] + i * M for i in range(k)])

    def forward(self, observed, goals, batch_split, prediction_truth=None, n_predict=None):
        """reference vae/vae.py:188-315 -> (rel_pred_scene, pred_scene, z_distr_xy, z_distr_x): two lists over the modes of
        [T_obs-1 + T_pred-1, M, 5] / [.., M, 2] tensors, the latent statistics [M, 2 latent_dim] (training) or None, and None
        (``desire``: the prior is the standard normal; ``vae_encoder_x`` stays unused as in the reference)."""
        assert ((prediction_truth is None) + (n_predict is None)) == 1
        if not self.desire:
            raise NotImplementedError('VAE.desire = False (latent prior from vae_encoder_x) is not wired to a trainer option in '
                                      'the reference either (vae/vae.py:81 fixes it to True)')
        from ..lstm.non_gridbased_pooling import _StatefulInteractionEncoder
        if isinstance(self.pool, _StatefulInteractionEncoder):
            # NearestNeighborLSTM / TrajectronPooling keep an LSTM state of their own.  The reference resets it ONCE per forward
            # (vae/vae.py:229-231) and carries it through the observation encoder, the prediction encoder and then every decoder
            # mode in turn; the sequence driver here starts each of those runs from a zero interaction-encoder state, which
            # would silently compute something else.  Not supported rather than different.
            raise NotImplementedError('VAE with a stateful interaction module (%s): the reference threads pool_lstm\'s state '
                                      'through obs_encoder -> pred_encoder -> every decoder mode; use a scene-local module '
                                      '(grid pooling, nn, hiddenstatemlp, attentionmlp)' % type(self.pool).__name__)
        obs_run, pred_run = self._runners()
        dev = self.obs_encoder.weight_ih.device
        if dev.type != 'cuda':
            raise RuntimeError('VAE parameters live on %s: move the model to a ROCm device; the MI355X path has no CPU fallback' % dev)
        observed = _lib.f32c(observed, dev)
        M, H, K = observed.size(1), self.hidden_dim, int(self.num_modes)
        if prediction_truth is not None:
            if isinstance(prediction_truth, (list, tuple)):
                prediction_truth = torch.stack(list(prediction_truth), dim=0)
            truth = _lib.f32c(prediction_truth, dev)
            T_dec = truth.size(0)
        else:
            truth, T_dec = None, n_predict - 1
        goals_d = _lib.f32c(goals, dev) if goals is not None else None
        grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        z_distr_xy = None
        if self.training:
            # latent statistics from the observation encoder's and the prediction encoder's final hidden states
            # (vae/vae.py:253-277).  prediction_truth is required in training mode, as in the reference (:255)
            assert truth is not None
            frames = torch.cat([observed[-1:], truth], dim=0)
            if grad:
                from ..lstm.training import run_sequence_with_grad
                _, _, h_obs = run_sequence_with_grad(obs_run, observed, goals_d, batch_split, None, 0)
                _, _, h_pred = run_sequence_with_grad(pred_run, frames, goals_d, batch_split, None, 0)
            else:
                _, _, h_obs = obs_run._run_sequence(observed, goals_d, batch_split, None, 0, want_h_final=True)
                _, _, h_pred = pred_run._run_sequence(frames, goals_d, batch_split, None, 0, want_h_final=True)
            z_mu, z_var_log = self.vae_encoder_xy(torch.cat([h_obs, h_pred], dim=1))
            z_distr_xy = torch.cat((z_mu, z_var_log), dim=1)
            # reparametrisation: one host draw per mode, in the order the reference's mode loop draws them (:93-94)
            scales = []
            for _ in range(K):
                epsilon = torch.empty(size=(M, self.latent_dim)).normal_(mean=0, std=1).to(dev)
                scales.append(self.vae_decoder(z_mu + torch.exp(0.5 * z_var_log) * epsilon))
        else:
            # evaluation: samples of the prior N(0, exp(1)) -- z_mu_obs = 0, z_var_log_obs = 1 (:273-274, :97-98)
            z_mu_obs, z_var_log_obs = torch.zeros(M, self.latent_dim), torch.ones(M, self.latent_dim)
            scales = [self.vae_decoder(sample_multivariate_distribution(z_mu_obs, z_var_log_obs).to(dev)) for _ in range(K)]
        # the K decoder passes: one sequence over K replicas of the scenes when the interaction module is scene-local
        from ..sgan.sgan import _scene_local
        if K > 1 and _scene_local(self.pool):
            runs = [(observed.repeat(1, K, 1), goals_d.repeat(K, 1) if goals_d is not None else None,
                     self._replicate(batch_split, K), truth.repeat(1, K, 1) if truth is not None else None,
                     torch.cat(scales, dim=0), K)]
        else:
            runs = [(observed, goals_d, batch_split, truth, sc, 1) for sc in scales]
        rel_list, pred_list = [], []
        for obs_k, goals_k, split_k, truth_k, scale_k, reps in runs:
            # the decoder consumes prediction_truth[:-1] + the frame fed back: T_dec decoder steps (vae/vae.py:283-301)
            if grad:
                from ..lstm.training import run_sequence_with_grad
                rel, pred, _ = run_sequence_with_grad(obs_run, obs_k, goals_k, split_k, truth_k, T_dec, {'h_scale': scale_k})
            else:
                rel, pred, _ = obs_run._run_sequence(obs_k, goals_k, split_k, truth_k, T_dec, h_scale=scale_k.detach())
            rel_list.extend(rel.chunk(reps, dim=1))
            pred_list.extend(pred.chunk(reps, dim=1))
        return rel_list, pred_list, z_distr_xy, None


class VAEPredictor(object):
    """reference vae/vae.py:346-398"""

    def __init__(self, model):
        self.model = model

    def save(self, state, filename):
        with open(filename, 'wb') as f:
            torch.save(self, f)
        with open(filename + '.state', 'wb') as f:
            torch.save(state, f)

    @staticmethod
    def load(filename):
        with open(filename, 'rb') as f:
            return torch.load(f, weights_only=False)

    def __call__(self, paths, scene_goal, n_predict=12, modes=1, predict_all=True, obs_length=9, start_length=0, args=None):
        self.model.eval()
        self.model.num_modes = modes
        with torch.no_grad():
            xy = trajdata.paths_to_xy(paths)
            batch_split = [0, xy.shape[1]]
            normalize = bool(getattr(args, 'normalize_scene', False))
            if normalize:
                xy, rotation, center, scene_goal = trajdata.center_scene(xy, obs_length, goals=scene_goal)
            xy = torch.tensor(np.asarray(xy), dtype=torch.float32)
            scene_goal = torch.tensor(np.asarray(scene_goal), dtype=torch.float32)
            batch_split = torch.tensor(batch_split, dtype=torch.int64)
            multimodal_outputs = {}
            _, output_scenes_list, _, _ = self.model(xy[start_length:obs_length], scene_goal, batch_split, n_predict=n_predict)
            for num_p, output_scenes in enumerate(output_scenes_list):
                output_scenes = output_scenes.cpu().numpy()
                if normalize:
                    output_scenes = trajdata.inverse_scene(output_scenes, rotation, center)
                output_primary = output_scenes[-n_predict:, 0]
                output_neighs = output_scenes[-n_predict:, 1:]
                multimodal_outputs[num_p] = [output_primary, output_neighs if num_p == 0 else []]
        return multimodal_outputs
