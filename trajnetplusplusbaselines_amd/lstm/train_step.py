"""One optimisation step as the reference's Trainer.train_batch does it (lstm/trainer.py:229-269), plus its
data-parallel form: scenes sharded over ranks, one bucketed gradient all-reduce (RCCL over xGMI on ROCm) per step."""
import torch

from .. import parallel


def train_batch(model, optimizer, criterion, batch_scene, batch_scene_goal, batch_split, obs_length=9, pred_length=12,
                batch_size=None, group=None, n_global_scenes=None):
    """batch_scene [obs+pred, M, 2] (NaN = absent), batch_scene_goal [M, 2], batch_split [B+1].

    Single process: loss = criterion(rel_outputs[-pred_length:], targets, batch_split) * batch_size, backward, step
    (lstm/trainer.py:252-267).  With a process group, `batch_scene` is this rank's shard of scenes: the loss is scaled
    so that SUM-reduced gradients equal the single-process gradient (parallel.scale_loss_for_sharding), gradients are
    all-reduced in one flattened bucket, and every rank applies the same optimizer step.  Returns the loss value."""
    model.train()
    dev = next(model.parameters()).device
    batch_scene = batch_scene.to(dev)
    split = torch.as_tensor(batch_split, dtype=torch.int64)
    n_local = split.numel() - 1
    batch_size = batch_size or n_local
    observed = batch_scene[:obs_length].clone()
    prediction_truth = batch_scene[obs_length:-1].clone()
    targets = batch_scene[obs_length:obs_length + pred_length] - batch_scene[obs_length - 1:obs_length + pred_length - 1]
    rel_outputs, outputs = model(observed, batch_scene_goal, split, prediction_truth)
    loss_mean = criterion(rel_outputs[-pred_length:], targets, split, outputs[-pred_length:]) \
        if getattr(criterion, 'col_wt', 0) else criterion(rel_outputs[-pred_length:], targets, split)
    if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
        n_global = n_global_scenes if n_global_scenes is not None else n_local * torch.distributed.get_world_size(group)
        loss = parallel.scale_loss_for_sharding(loss_mean, batch_size, n_local, n_global)
    else:
        loss = loss_mean * batch_size
    optimizer.zero_grad()
    loss.backward()
    parallel.allreduce_gradients(list(model.parameters()), group=group)
    optimizer.step()
    return float(loss.detach())
