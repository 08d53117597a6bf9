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
                obs_length=9, pred_length=12, start_length=0, group=None, n_global_scenes=None, pad_to=None):
    """batch_scene [obs+pred, M, 2], step_type 'g' | 'd' (reference sgan/trainer.py:258-300).  `criterion` must keep the
    batch dimension (PredictionLoss(keep_batch_dim=True)), as the reference's top-k loss needs per-scene values.

    With a process group, `batch_scene` is this rank's shard of scenes (``parallel.shard_batch``; pass its
    ``n_global_scenes`` and ``pad_to``): the variety loss is a SUM over scenes and enters as it is, the adversarial BCE
    terms are MEANS over scenes and enter with weight n_local / n_global, the updated network's gradients are SUM-reduced
    over the ranks (one flattened bucket), and every rank applies the same optimiser step.  The noise vectors and the noisy
    labels come from the ranks' own `torch` / `random` generators: seed them identically on every rank (the reference draws
    ONE noise vector per sample and ONE label per batch) -- ranks that run the same code stay in step without communication.
    Returns this rank's loss value."""
    if not model.training:
        model.train()
    dev = next(model.parameters()).device
    distributed = group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                        and torch.distributed.get_world_size() > 1)
    split = torch.as_tensor(batch_split, dtype=torch.int64)
    n_local = split.numel() - 1
    if n_local <= 0:
        raise ValueError('empty shard: give every rank at least one scene (parallel.shard_batch does when scenes >= ranks)')
    batch_scene = batch_scene.to(dev)
    seq_length = obs_length + pred_length
    observed = batch_scene[start_length:obs_length].clone()
    prediction_truth = batch_scene[obs_length:].clone()
    targets = batch_scene[obs_length:seq_length] - batch_scene[obs_length - 1:seq_length - 1]
    rel_output_list, outputs, scores_real, scores_fake = model(observed, batch_scene_goal, batch_split, prediction_truth,
                                                               step_type=step_type, pred_length=pred_length, pad_to=pad_to)
    if distributed:
        n_global = n_global_scenes if n_global_scenes is not None else n_local * torch.distributed.get_world_size(group)
        w = float(n_local) / float(n_global)
        if step_type == 'd':
            loss = gan_d_loss(scores_real, scores_fake) * w
        else:
            loss = variety_loss(criterion, rel_output_list, targets, batch_split, pred_length)
            if model.d_steps:
                loss = loss + gan_g_loss(scores_fake) * w
    else:
        loss = loss_criterion(model, criterion, rel_output_list, targets, batch_split, scores_fake, scores_real, step_type,
                              pred_length)
    read_back = _LossReadBack(loss)      # the value is final here: it travels to the host while backward + update run
    opt = g_optimizer if step_type == 'g' else d_optimizer
    opt.zero_grad()                      # only the updated network's, as the reference does (sgan/trainer.py:286-294)
    loss.backward()
    if distributed:
        from .. import parallel
        net = model.generator if step_type == 'g' else model.discriminator
        parallel.allreduce_gradients(list(net.parameters()), group=group)
    opt.step()
    return read_back.value()
