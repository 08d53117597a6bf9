"""S-GAN on MI355X (reference sgan/sgan.py:46-630): the generator is the LSTM forecaster of lstm/lstm.py with a noise
interface between encoder and decoder, the discriminator an encoder LSTM over observed + predicted frames followed
by a small MLP on the primaries' hidden state.  Both run on the same HIP sequence driver (tnp_lstm_forward_ex);
class names, constructor arguments, state_dict keys and return values mirror the reference.  Forward and training: generator
and discriminator train through every interaction module of the reference (grid and non-grid), incl. the generator step
that back-propagates from the scores through the discriminator's interaction module into the predicted positions
(lstm/training.py: tnp_pool_*_pos_backward / tnp_directional_scatter_backward) -- the reference's trainer gives the
discriminator a deep copy of the generator's module (sgan/trainer.py:566-592)."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .. import data as trajdata
from ..lstm.lstm import LSTM, drop_distant  # noqa: F401  (drop_distant is re-exported like the reference does)


def get_noise(shape, noise_type, device):
    """reference sgan/sgan.py:27-32"""
    if noise_type == 'gaussian':
        return torch.randn(*shape, device=device)
    if noise_type == 'uniform':
        return torch.rand(*shape, device=device).sub_(0.5).mul_(2.0)
    raise ValueError('Unrecognized noise type "%s"' % noise_type)


def make_mlp(dim_list, activation='relu', batch_norm=True, dropout=0):
    """reference sgan/sgan.py:34-44 (a ReLU follows EVERY Linear, including the last one)"""
    layers = []
    for dim_in, dim_out in zip(dim_list[:-1], dim_list[1:]):
        layers.append(nn.Linear(dim_in, dim_out))
        if activation == 'relu':
            layers.append(nn.ReLU())
        elif activation == 'leakyrelu':
            layers.append(nn.LeakyReLU())
        if dropout > 0:
            layers.append(nn.Dropout(p=dropout))
    return nn.Sequential(*layers)


class LSTMGenerator(LSTM):
    def __init__(self, embedding_dim=64, hidden_dim=128, pool=None, pool_to_input=True, goal_dim=None, goal_flag=False,
                 noise_dim=8, no_noise=False, noise_type='gaussian'):
        """Arguments as in reference sgan/sgan.py:135-153."""
        super(LSTMGenerator, self).__init__(embedding_dim=embedding_dim, hidden_dim=hidden_dim, pool=pool,
                                            pool_to_input=pool_to_input, goal_dim=goal_dim, goal_flag=goal_flag)
        self.noise_dim = noise_dim
        self.no_noise = no_noise
        self.noise_type = noise_type
        self.mlp_decoder_context = make_mlp([self.hidden_dim, self.hidden_dim - self.noise_dim])
        #: slots the reference would pad the scenes to when this call holds a SHARD of a larger batch (_lib.SceneIndex);
        #: None = the largest scene of the call.  Set by SGAN.forward(pad_to=...) / sgan.train_step for sharded training
        self.pad_to = None

    def forward(self, observed, goals, batch_split, prediction_truth=None, n_predict=None, noise=None):
        """reference sgan/sgan.py:302-399.  `noise` ([noise_dim]) may be given explicitly; by default it is drawn with
        get_noise() on the host exactly where the reference draws it."""
        assert ((prediction_truth is None) + (n_predict is None)) == 1
        if prediction_truth is not None:
            if isinstance(prediction_truth, (list, tuple)):
                prediction_truth = torch.stack(list(prediction_truth), dim=0)
            truth = prediction_truth[:-1]            # the reference feeds prediction_truth[:-1] (:362-364)
            T_dec = truth.size(0)
        else:
            truth, T_dec = None, n_predict - 1
        training = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.no_noise:
            if training:
                from ..lstm.training import run_sequence_with_grad
                rel, pred, _ = run_sequence_with_grad(self, observed, goals, batch_split, truth, T_dec, {'pad_to': self._pad(batch_split)})
                return rel, pred
            rel, pred, _ = self._run_sequence(observed, goals, batch_split, truth, T_dec, pad_to=self._pad(batch_split))
            return rel, pred
        if noise is None:
            noise = get_noise((self.noise_dim,), self.noise_type, device='cpu')
        if training:   # autograd through the sequence incl. the noise / context MLP (lstm/training.py)
            from ..lstm.training import run_sequence_with_grad
            rel, pred, _ = run_sequence_with_grad(self, observed, goals, batch_split, truth, T_dec,
                                                  {'noise': noise, 'pad_to': self._pad(batch_split)})
            return rel, pred
        lin = self.mlp_decoder_context[0]
        rel, pred, _ = self._run_sequence(observed, goals, batch_split, truth, T_dec, w_ctx=lin.weight, b_ctx=lin.bias,
                                          noise=noise, pad_to=self._pad(batch_split))
        return rel, pred

    def _pad(self, batch_split):
        return getattr(self, 'pad_to', None)

    def sample_k(self, observed, goals, batch_split, prediction_truth=None, n_predict=None, k=1):
        """k independent samples (the k `generator(...)` calls of reference sgan/sgan.py:94-100) as ONE sequence over k
        replicas of the scenes, replica i carrying noise vector i: scenes never interact, so the results equal the
        one-by-one calls, with k times the tracks per kernel launch.  Noise is drawn in the order the k calls draw it."""
        if k == 1 or self.no_noise or not _scene_local(self.pool):
            outs = [self.forward(observed, goals, batch_split, prediction_truth, n_predict) for _ in range(k)]
            return [o[0] for o in outs], [o[1] for o in outs]
        noise = torch.stack([get_noise((self.noise_dim,), self.noise_type, device='cpu') for _ in range(k)], dim=0)
        if isinstance(prediction_truth, (list, tuple)):
            prediction_truth = torch.stack(list(prediction_truth), dim=0)
        rel, pred = self.forward(observed.repeat(1, k, 1), goals.repeat(k, 1) if goals is not None else None,
                                 _replicate_scenes(batch_split, k),
                                 prediction_truth.repeat(1, k, 1) if prediction_truth is not None else None, n_predict,
                                 noise=noise)
        return list(rel.chunk(k, dim=1)), list(pred.chunk(k, dim=1))


def _replicate_scenes(batch_split, k):
    """batch_split of k copies of the batch laid side by side along the track axis"""
    split = torch.as_tensor(batch_split, dtype=torch.int64).cpu()
    M = int(split[-1])
    return torch.cat([split[:1]] + [split[1:] + i * M for i in range(k)])


def _scene_local(pool):
    """interaction modules whose output for a scene does not depend on the other scenes of the batch (TrajectronPooling
    sums over the whole batch, a reference quirk) -- only those may be batched as replicated scenes"""
    return pool is None or type(pool).__name__ == 'GridBasedPooling'


class LSTMDiscriminator(LSTM):
    _ENCODER_ONLY = True

    def __init__(self, embedding_dim=64, hidden_dim=128, pool=None, pool_to_input=True, goal_dim=None, goal_flag=False):
        """Arguments as in reference sgan/sgan.py:402-446.  Only `encoder` is used: the base class constructs neither decoder
        nor hidden2normal here, so the state_dict AND the order in which parameters are drawn from the RNG match the
        reference (a seeded default init gives the reference's weights, tests/golden/sgan_full.npz)."""
        super(LSTMDiscriminator, self).__init__(embedding_dim=embedding_dim, hidden_dim=hidden_dim, pool=pool,
                                                pool_to_input=pool_to_input, goal_dim=goal_dim, goal_flag=goal_flag)
        self.real_classifier = make_mlp([self.hidden_dim, int(self.hidden_dim / 2), int(self.hidden_dim / 4), 1])
        self._dummy_head = None
        self.pad_to = None            # see LSTMGenerator.pad_to

    # the C descriptor wants a decoder cell and a Hidden2Normal head; an encoder-only run never uses the first and
    # discards the output of the second, so hand it the encoder and a zero head
    def _decoder_cell(self):
        return self.encoder

    def _normal_head(self):
        dev = self.encoder.weight_ih.device
        if self._dummy_head is None or self._dummy_head[0].device != dev:
            self._dummy_head = (torch.zeros(5, self.hidden_dim, device=dev), torch.zeros(5, device=dev))
        return self._dummy_head

    def forward(self, observed, prediction, goals, batch_split):
        """[T_obs,M,2], [T_pred,M,2] -> scores [B,1] of the primaries (reference sgan/sgan.py:512-576)."""
        dev = self.encoder.weight_ih.device
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: gradients for the discriminator's parameters and, when the scored prediction comes from the
            # generator, for the positions themselves (lstm/training.py, opts input_grad)
            from ..lstm.training import run_sequence_with_grad
            frames = torch.cat([observed.to(dev, torch.float32), prediction.to(dev, torch.float32)], dim=0)
            _, _, h = run_sequence_with_grad(self, frames, goals, batch_split, None, 0,
                                             {'input_grad': True, 'pad_to': getattr(self, 'pad_to', None)})
            prim = _lib.SceneIndex.get(batch_split, dev).starts[:-1].long()   # cached on the device: no host sync
            return self.real_classifier(h[prim])
        frames = torch.cat([_lib.f32c(observed, dev), _lib.f32c(prediction, dev)], dim=0)
        _, _, h = self._run_sequence(frames, goals, batch_split, None, 0, want_h_final=True, pad_to=getattr(self, 'pad_to', None))
        x = h[_lib.SceneIndex.get(batch_split, dev).starts[:-1].long()]
        for layer in self.real_classifier:
            if isinstance(layer, nn.Linear):
                x = _lib.linear_forward(x, layer.weight.detach(), layer.bias.detach(), relu=True)
        return x

    def score_pair(self, observed, real, fake, goals, batch_split):
        """(scores_real, scores_fake) = (self(observed, real, ...), self(observed, fake, ...)) of reference
        sgan/sgan.py:109-110 as one encoder run over 2 replicas of the scenes (the classifier acts row by row)."""
        if real.shape != fake.shape or not _scene_local(self.pool):
            return self.forward(observed, real, goals, batch_split), self.forward(observed, fake, goals, batch_split)
        dev = self.encoder.weight_ih.device
        obs = observed.to(dev, torch.float32)
        pred2 = torch.cat([real.to(dev, torch.float32), fake.to(dev, torch.float32)], dim=1)
        scores = self.forward(obs.repeat(1, 2, 1), pred2, goals.repeat(2, 1) if goals is not None else None,
                              _replicate_scenes(batch_split, 2))
        B = scores.shape[0] // 2
        return scores[:B], scores[B:]


class SGAN(torch.nn.Module):
    def __init__(self, generator=None, discriminator=None, k=1, d_steps=1, g_steps=1):
        """reference sgan/sgan.py:46-76"""
        super(SGAN, self).__init__()
        self.generator = generator if generator is not None else LSTMGenerator()
        self.g_steps = g_steps
        self.discriminator = discriminator if discriminator is not None else LSTMDiscriminator()
        self.d_steps = d_steps
        self.k = k
        #: run the k generator samples / the real + fake discriminator passes as replicated scenes of one sequence
        self.batch_samples = True
        #: 'd'-type steps run the generator without an autograd graph (its gradients are never applied on such a step)
        self.skip_generator_graph_on_d = True

    def forward(self, observed, goals, batch_split, prediction_truth=None, n_predict=None, step_type='g',
                pred_length=12, pad_to=None):
        """reference sgan/sgan.py:78-132: k generator samples, then real / fake discriminator scores.  ``pad_to``
        (extension): the batch-wide slot count when this call holds a shard of a larger batch (parallel.shard_batch)."""
        self.generator.pad_to = self.discriminator.pad_to = pad_to
        n_samples = 1 if step_type == 'd' else self.k           # the reference breaks out of its loop on a 'd' step
        # a discriminator step only updates the discriminator (sgan/trainer.py:286-300 steps d_optimizer): its generator
        # sample needs no autograd graph (the reference builds one and throws the generator's gradients away)
        with torch.set_grad_enabled(torch.is_grad_enabled() and not (step_type == 'd' and self.skip_generator_graph_on_d)):
            if self.batch_samples:
                rel_pred_list, pred_list = self.generator.sample_k(observed, goals, batch_split, prediction_truth,
                                                                   n_predict, n_samples)
            else:
                rel_pred_list, pred_list = [], []
                for _ in range(n_samples):
                    rel_pred_scene, pred_scene = self.generator(observed, goals, batch_split, prediction_truth, n_predict)
                    rel_pred_list.append(rel_pred_scene)
                    pred_list.append(pred_scene)
        pred_scene = pred_list[-1]
        if self.d_steps and (prediction_truth is not None):
            if isinstance(prediction_truth, (list, tuple)):
                prediction_truth = torch.stack(list(prediction_truth), dim=0)
            if self.batch_samples:
                scores_real, scores_fake = self.discriminator.score_pair(observed, prediction_truth,
                                                                         pred_scene[-pred_length:], goals, batch_split)
            else:
                scores_real = self.discriminator(observed, prediction_truth, goals, batch_split)
                scores_fake = self.discriminator(observed, pred_scene[-pred_length:], goals, batch_split)
            return rel_pred_list, pred_list, scores_real, scores_fake
        return rel_pred_list, pred_list, None, None


class SGANPredictor(object):
    """reference sgan/sgan.py:578-630"""

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

    def __call__(self, paths, scene_goal, n_predict=12, modes=1, predict_all=True, obs_length=9, start_length=0,
                 args=None):
        self.model.eval()
        self.model.d_steps = 0
        if modes is not None:
            self.model.k = modes
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
            _, output_scenes_list, _, _ = self.model(xy[:obs_length], scene_goal, batch_split, n_predict=n_predict)
            for num_p, output_scenes in enumerate(output_scenes_list):
                output_scenes = output_scenes.cpu().numpy()
                if normalize:
                    output_scenes = trajdata.inverse_scene(output_scenes, rotation, center)
                output_primary = output_scenes[-n_predict:, 0]
                output_neighs = output_scenes[-n_predict:, 1:]
                multimodal_outputs[num_p] = [output_primary, output_neighs if num_p == 0 else []]
        return multimodal_outputs
