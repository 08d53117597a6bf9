"""One optimisation step of the reference's VAE trainer (vae/trainer.py:229-280) on the HIP path."""
import random

import torch

from .loss import KLDLoss


def batch_loss(model, criterion, batch_scene, batch_scene_goal, batch_split, obs_length=9, pred_length=12, batch_size=None,
               alpha_kld=1.0, start_length=0, kld=None):
    """(loss, reconstruction loss): reconstruction = mean over the modes of criterion(rel_outputs[-pred_length:], targets) *
    batch_size, loss = reconstruction + alpha_kld * KLD(z_distr_xy) * batch_size (vae/trainer.py:257-274)."""
    dev = next(model.parameters()).device
    split = torch.as_tensor(batch_split, dtype=torch.int64)
    batch_size = batch_size or (split.numel() - 1)
    seq_length = obs_length + pred_length
    scene = batch_scene.to(dev, torch.float32)
    observed = scene[start_length:obs_length]
    prediction_truth = scene[obs_length:seq_length - 1]
    targets = scene[obs_length:seq_length] - scene[obs_length - 1:seq_length - 1]
    rel_outputs, _, z_distr_xy, z_distr_x = model(observed, batch_scene_goal, split, prediction_truth)
    reconstr_loss = 0
    for rel_outputs_mode in rel_outputs:
        reconstr_loss = reconstr_loss + criterion(rel_outputs_mode[-pred_length:], targets, split) * batch_size
    reconstr_loss = reconstr_loss / model.num_modes
    kld_loss = (kld or KLDLoss())(z_distr_xy, split, z_distr_x) * batch_size
    return reconstr_loss + alpha_kld * kld_loss, reconstr_loss


def train_batch(model, optimizer, criterion, batch_scene, batch_scene_goal, batch_split, obs_length=9, pred_length=12,
                batch_size=None, alpha_kld=1.0, start_length=0, obs_dropout=False):
    """Trainer.train_batch of vae/trainer.py:229-280; returns the reconstruction loss like the reference does."""
    if not model.training:
        model.train()
    if obs_dropout:
        start_length = random.randint(0, obs_length - 2)
    loss, reconstr = batch_loss(model, criterion, batch_scene, batch_scene_goal, batch_split, obs_length, pred_length, batch_size,
                                alpha_kld, start_length)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(reconstr.detach())
