"""One optimisation step as the reference's Trainer.train_batch does it (lstm/trainer.py:229-269), plus its
data-parallel form: scenes sharded over ranks, gradients all-reduced (RCCL over xGMI on ROCm) once per step."""
import random

import torch

from .. import parallel
from .loss import L2Loss, PredictionLoss


def train_batch(model, optimizer, criterion, batch_scene, batch_scene_goal, batch_split, obs_length=9, pred_length=12,
                batch_size=None, group=None, n_global_scenes=None, pad_to=None, start_length=0, obs_dropout=False,
                buckets=None, overlap=False):
    """batch_scene [obs+pred, M, 2] (NaN = absent), batch_scene_goal [M, 2], batch_split [B+1].

    Single process: loss = criterion(rel_outputs[-pred_length:], targets, batch_split, primary_prediction) * batch_size,
    backward, step (lstm/trainer.py:246-267; ``primary_prediction`` = the truth frames with the primaries' columns
    replaced by the model's predictions, :258-260 -- what the auxiliary collision loss is evaluated on;
    ``obs_dropout`` draws ``start_length`` like :246-247).

    With a process group, `batch_scene` is this rank's shard of scenes (``parallel.shard_batch``): pass its
    ``n_global_scenes`` and ``pad_to`` (the largest scene of the whole batch, so that padded-slot effects equal the
    unsharded batch's).  The loss is scaled so that SUM-reduced gradients equal the single-process gradient
    (parallel.scale_loss_for_sharding), gradients are all-reduced -- through ``buckets`` (parallel.GradBuckets: flat
    persistent buffers, asynchronous launch) when given, else one flattened bucket -- and every rank applies the same
    optimizer step.  A rank whose shard is empty (fewer scenes than ranks) back-propagates a zero-weighted one-track
    dummy scene: it contributes zero gradients for exactly the parameters the other ranks have gradients for, so the
    collectives line up.  ``overlap=True`` (our ``LSTM`` only) all-reduces every gradient from inside the backward pass
    as soon as it is enqueued (parallel.GradReducer) instead of after it.  Returns the loss value of this rank."""
    if not model.training:       # (a no-op walk over every sub-module otherwise: 0.1 ms of host time in front of the first kernel)
        model.train()
    dev = next(model.parameters()).device
    distributed = group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())
    split = torch.as_tensor(batch_split, dtype=torch.int64)
    n_local = split.numel() - 1
    batch_size = batch_size or n_local
    if obs_dropout:
        start_length = random.randint(0, obs_length - 2)
        if distributed and torch.distributed.get_world_size(group) > 1:
            # the reference draws ONE start_length per batch (lstm/trainer.py:246-247): the shards of a batch must be
            # encoded from the same observation window, whatever the ranks' own `random` streams hold
            box = [start_length]
            src = torch.distributed.get_global_rank(group, 0) if group is not None else 0
            torch.distributed.broadcast_object_list(box, src=src, group=group)
            start_length = int(box[0])
    empty = n_local <= 0 or int(split[-1]) <= 0
    if empty and not distributed:
        raise ValueError('empty batch')
    if empty:
        t = torch.arange(obs_length + pred_length, dtype=torch.float32).view(-1, 1, 1)
        batch_scene = (t * torch.tensor([0.1, 0.05])).expand(-1, 1, 2).contiguous()
        batch_scene_goal, split, n_local = torch.zeros(1, 2), torch.tensor([0, 1]), 0
    batch_scene = batch_scene.to(dev)
    # (the reference clones both slices, lstm/trainer.py:251-252; its forward pass writes into neither and neither does
    # ours -- two copies less in front of the first kernel)
    observed = batch_scene[start_length:obs_length]
    prediction_truth = batch_scene[obs_length:obs_length + pred_length - 1]
    targets = batch_scene[obs_length:obs_length + pred_length] - batch_scene[obs_length - 1:obs_length + pred_length - 1]
    in_backward = distributed and overlap and hasattr(model, '_grad_reduce_fn')
    if distributed and n_global_scenes is None and empty:
        raise ValueError('an empty shard needs n_global_scenes')
    if distributed:
        n_global = n_global_scenes if n_global_scenes is not None else n_local * torch.distributed.get_world_size(group)
    if in_backward:
        model._grad_reduce_fn = parallel.GradReducer(group)
    try:
        rel_outputs, outputs = model(observed, batch_scene_goal, split, prediction_truth, pad_to=pad_to)
        loss = batch_loss(criterion, rel_outputs, outputs, batch_scene, targets, split, pred_length, batch_size,
                          shard=(n_local, n_global) if distributed else None)
        # The loss value leaves for the host NOW (pinned buffer + event), not through a blocking read after optimizer.step():
        # it is final once the forward pass is, so the caller gets its float back when the HOST has queued the step and
        # prepares the next batch while the GPU still runs this step's backward pass and optimiser update -- the reference's
        # loss.item() at the end of the step (lstm/trainer.py:268) leaves the GPU idle for that long at every step.
        read_back = _LossReadBack(loss)
        # gradients are cleared between the forward pass and backward(), where the reference does it (lstm/trainer.py:264):
        # by then the forward kernels are queued and the host's time is free
        if buckets is not None:
            buckets.zero()
        else:
            optimizer.zero_grad()
        # backward on THIS thread: torch's engine otherwise hands the device's node queue to a worker thread, whose wake-up and
        # GIL hand-over cost 0.3 ms of a batch_size-8 step (2.59 -> 2.29 ms; the backward is one native sweep either way)
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
    finally:
        if in_backward:
            # also on an exception (NaN check, out of memory ...): a reducer left attached would all-reduce from inside
            # the next, unrelated backward pass of this model
            model._grad_reduce_fn = None              # gradients were summed over the ranks inside the backward pass
    if in_backward:
        pass
    elif buckets is not None:
        buckets.launch_all()
        buckets.wait()
    elif distributed:
        parallel.allreduce_gradients(list(model.parameters()), group=group)
    optimizer.step()
    return read_back.value()


class _LossReadBack(object):
    """float(loss) without waiting for work queued after the loss: an asynchronous copy into pinned host memory and an event.
    (The pinned scalar comes from torch's caching host allocator: a few microseconds after the first call.)"""

    def __init__(self, loss):
        self.loss, self.event, self.buf = loss.detach(), None, None
        if self.loss.is_cuda:
            with torch.cuda.device(self.loss.device):
                self.buf = torch.empty((), dtype=torch.float32, pin_memory=True)
                self.buf.copy_(self.loss.float(), non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record()

    def value(self):
        if self.event is None:
            return float(self.loss)
        self.event.synchronize()
        return float(self.buf)


def batch_loss(criterion, rel_outputs, outputs, batch_scene, targets, split, pred_length, batch_size, shard=None):
    """The scalar `train_batch` back-propagates.  Single process (``shard`` None): ``criterion(...) * batch_size`` with
    ``primary_prediction`` handed over when the criterion carries a collision weight (lstm/trainer.py:258-264).

    ``shard = (n_local_scenes, n_global_scenes)``: this rank holds a shard of the batch and gradients are SUM-reduced.
    The criterion has two kinds of terms: the primaries' NLL / L2 is a MEAN over (frames x scenes) (lstm/loss.py:85-91),
    so a shard contributes ``mean_local * n_local / n_global``; the collision penalty is a SUM over the scenes
    (lstm/loss.py:147-161, added un-normalised at :90), so a shard contributes its own sum as it is.  Criteria that
    expose ``terms()`` (ours do) are scaled term by term; any other criterion is treated as a mean."""
    dev = rel_outputs.device
    positions = None
    if getattr(criterion, 'col_wt', 0):
        prim = split[:-1].to(dev)
        positions = batch_scene[-pred_length:].clone()
        positions[:, prim] = outputs[-pred_length:, prim]
    if shard is None and isinstance(criterion, (PredictionLoss, L2Loss)) and not criterion.keep_batch_dim:
        # our criteria take the slice and the factor inside (same value to an ulp; three autograd nodes and five launches less
        # per step: slice -> zero fill + copy of its gradient, multiplication forward and backward)
        loss, col = criterion.terms(rel_outputs, targets, split, positions, tail=pred_length, times=float(batch_size))
        return loss if col is None else loss + col
    args = (rel_outputs[-pred_length:], targets, split) + ((positions,) if positions is not None else ())
    if shard is None:
        return criterion(*args) * batch_size
    n_local, n_global = shard
    if hasattr(criterion, 'terms'):
        mean_term, sum_term = criterion.terms(*args)
        loss = parallel.scale_loss_for_sharding(mean_term, batch_size, n_local, n_global)     # n_local == 0: weight 0
        if sum_term is not None:
            loss = loss + sum_term * (batch_size if n_local > 0 else 0.0)
        return loss
    return parallel.scale_loss_for_sharding(criterion(*args), batch_size, n_local, n_global)


def val_batch(model, criterion, batch_scene, batch_scene_goal, batch_split, obs_length=9, pred_length=12, batch_size=None,
              pad_to=None, start_length=0):
    """Trainer.val_batch (lstm/trainer.py:271-311): the validation loss with the neighbours' ground truth provided
    (teacher forced) and without it (``n_predict``, the evaluation scenario), each ``criterion(rel_outputs[-pred_length:],
    targets, batch_split) * batch_size`` -- evaluated inside the two forward passes (``LSTM.forward_with_loss``).
    Returns (loss, loss_test) as floats."""
    dev = next(model.parameters()).device
    split = torch.as_tensor(batch_split, dtype=torch.int64)
    batch_size = batch_size or (split.numel() - 1)
    batch_scene = batch_scene.to(dev)
    observed = batch_scene[start_length:obs_length]
    prediction_truth = batch_scene[obs_length:obs_length + pred_length - 1].clone()
    targets = batch_scene[obs_length:obs_length + pred_length] - batch_scene[obs_length - 1:obs_length + pred_length - 1]
    was_training = model.training
    model.eval()
    with torch.no_grad():
        _, _, loss = model.forward_with_loss(observed, batch_scene_goal, split, targets, criterion,
                                             prediction_truth=prediction_truth, pad_to=pad_to)
        _, _, loss_test = model.forward_with_loss(observed.clone(), batch_scene_goal, split, targets, criterion,
                                                  n_predict=pred_length, pad_to=pad_to)
    model.train(was_training)
    return float(loss) * batch_size, float(loss_test) * batch_size
