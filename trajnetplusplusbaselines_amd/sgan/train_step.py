"""One S-GAN optimisation step on the MI355X path: mirror of ``Trainer.train_batch`` / ``loss_criterion`` /
``variety_loss`` of the reference (sgan/trainer.py:258-369)."""
import torch

from ..lstm.loss import gan_d_loss, gan_g_loss, variety_loss
from ..lstm.train_step import _LossReadBack


def loss_criterion(model, criterion, rel_output_list, targets, batch_split, scores_fake, scores_real, step_type,
                   pred_length=12):
    """reference sgan/trainer.py:330-369"""
    if step_type == 'd':
        return gan_d_loss(scores_real, scores_fake)
    loss = variety_loss(criterion, rel_output_list, targets, batch_split, pred_length)
    if model.d_steps:
        loss = loss + gan_g_loss(scores_fake)
    return loss


def train_batch(model, g_optimizer, d_optimizer, criterion, batch_scene, batch_scene_goal, batch_split, step_type,
                obs_length=9, pred_length=12, start_length=0):
    """batch_scene [obs+pred, M, 2], step_type 'g' | 'd' (reference sgan/trainer.py:258-300).  `criterion` must keep the
    batch dimension (PredictionLoss(keep_batch_dim=True)), as the reference's top-k loss needs per-scene values."""
    if not model.training:
        model.train()
    dev = next(model.parameters()).device
    batch_scene = batch_scene.to(dev)
    seq_length = obs_length + pred_length
    observed = batch_scene[start_length:obs_length].clone()
    prediction_truth = batch_scene[obs_length:].clone()
    targets = batch_scene[obs_length:seq_length] - batch_scene[obs_length - 1:seq_length - 1]
    rel_output_list, outputs, scores_real, scores_fake = model(observed, batch_scene_goal, batch_split, prediction_truth,
                                                               step_type=step_type, pred_length=pred_length)
    loss = loss_criterion(model, criterion, rel_output_list, targets, batch_split, scores_fake, scores_real, step_type,
                          pred_length)
    read_back = _LossReadBack(loss)      # the value is final here: it travels to the host while backward + update run
    opt = g_optimizer if step_type == 'g' else d_optimizer
    opt.zero_grad()
    loss.backward()
    opt.step()
    return read_back.value()
