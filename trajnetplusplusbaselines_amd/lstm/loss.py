"""Losses of the reference's lstm/loss.py (same class names and call signatures), evaluated on the primaries of a
batch by csrc/loss.hip: forward values and, on the training path, the analytic backward (tnp_primary_loss_backward)."""
import torch

from .. import _lib


class _PrimaryLossFn(torch.autograd.Function):
    """loss = tnp_primary_loss_forward(inputs, ...); d(loss)/d(inputs) by tnp_primary_loss_backward: two launches each
    instead of the ~60 elementwise kernels of the same expression written with tensor ops."""

    @staticmethod
    def forward(ctx, inputs, targets, batch_split, mode, background_rate, keep_batch_dim, scale, tail=None):
        dev = inputs.device
        full = _lib.f32c(inputs.detach())
        inp = full if tail is None else full[full.size(0) - tail:]      # the last `tail` frames (contiguous: leading dimension)
        tgt = _lib.f32c(targets.detach(), dev)
        T, M = inp.size(0), inp.size(1)
        idx = _lib.SceneIndex.get(batch_split, dev)
        out = torch.empty(idx.B if keep_batch_dim else 1, dtype=torch.float32, device=dev)
        ws = torch.empty(T * idx.B, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().tnp_primary_loss_forward(mode, _lib.ptr(inp), _lib.ptr(tgt), _lib.ptr(idx.starts), idx.B, T, M,
                                                       float(background_rate), int(keep_batch_dim), float(scale),
                                                       _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()),
                   'tnp_primary_loss_forward')
        ctx.args = (inp, tgt, idx, mode, float(background_rate), int(keep_batch_dim), float(scale), full.size(0))
        return out if keep_batch_dim else out[0]

    @staticmethod
    def backward(ctx, grad_out):
        inp, tgt, idx, mode, bg, keep, scale, frames = ctx.args
        T, M = inp.size(0), inp.size(1)
        g = _lib.f32c(grad_out.detach(), inp.device).reshape(-1)
        # gradient of ALL frames handed in: the kernel fills the last T, the frames in front of them (`tail`) get zeros -- what
        # slicing outside (inputs[-T:]) costs as a zero fill of everything plus a copy under autograd
        d_full = torch.empty(frames, M, 5, dtype=torch.float32, device=inp.device)
        if frames > T:
            d_full[:frames - T].zero_()
        d_inputs = d_full[frames - T:]
        _lib.check(_lib.lib().tnp_primary_loss_backward(mode, _lib.ptr(inp), _lib.ptr(tgt), _lib.ptr(idx.starts), idx.B, T, M, bg,
                                                        keep, scale, _lib.ptr(g), _lib.ptr(d_inputs), _lib.stream_ptr()),
                   'tnp_primary_loss_backward')
        return d_full, None, None, None, None, None, None, None


def _primary_loss(mode, inputs, targets, batch_split, background_rate, keep_batch_dim, scale, tail=None):
    """``tail``: evaluate on the last ``tail`` frames of ``inputs`` (== passing inputs[-tail:], without autograd's slice node)"""
    _lib.require_device(inputs, 'inputs')
    if inputs.requires_grad and torch.is_grad_enabled():
        return _PrimaryLossFn.apply(inputs, targets, batch_split, mode, background_rate, keep_batch_dim, scale, tail)
    dev = inputs.device
    inputs = _lib.f32c(inputs.detach())
    if tail is not None:
        inputs = inputs[inputs.size(0) - tail:]
    targets = _lib.f32c(targets.detach(), dev)
    T, M = inputs.size(0), inputs.size(1)
    idx = _lib.SceneIndex.get(batch_split, dev)
    out = torch.empty(idx.B if keep_batch_dim else 1, dtype=torch.float32, device=dev)
    ws = torch.empty(T * idx.B, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().tnp_primary_loss_forward(mode, _lib.ptr(inputs), _lib.ptr(targets), _lib.ptr(idx.starts), idx.B,
                                                   T, M, float(background_rate), int(keep_batch_dim), float(scale),
                                                   _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr()),
               'tnp_primary_loss_forward')
    return out if keep_batch_dim else out[0]


def _collision_forward(pred, idx, col_wt, col_distance):
    T, M, ld = pred.shape
    partial = torch.empty(idx.B, dtype=torch.float32, device=pred.device)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    _lib.check(_lib.lib().tnp_collision_loss_forward(_lib.ptr(pred), ld, _lib.ptr(idx.starts), idx.B, T, M, float(col_wt),
                                                     float(col_distance), _lib.ptr(partial), _lib.ptr(out),
                                                     _lib.stream_ptr()), 'tnp_collision_loss_forward')
    return out[0]


class _CollisionLossFn(torch.autograd.Function):
    """tnp_collision_loss_forward / tnp_collision_loss_backward: the penalty back-propagates into the primaries'
    predicted positions (the neighbours are detached, reference lstm/loss.py:155)."""

    @staticmethod
    def forward(ctx, predictions, batch_split, col_wt, col_distance):
        pred = _lib.f32c(predictions.detach())
        idx = _lib.SceneIndex.get(batch_split, pred.device)
        ctx.args = (pred, idx, float(col_wt), float(col_distance))
        return _collision_forward(pred, idx, col_wt, col_distance)

    @staticmethod
    def backward(ctx, grad_out):
        pred, idx, col_wt, col_distance = ctx.args
        T, M, ld = pred.shape
        g = _lib.f32c(grad_out.detach(), pred.device).reshape(-1)
        d_pred = torch.empty_like(pred)
        _lib.check(_lib.lib().tnp_collision_loss_backward(_lib.ptr(pred), ld, _lib.ptr(idx.starts), idx.B, T, M, col_wt,
                                                          col_distance, _lib.ptr(g), _lib.ptr(d_pred), _lib.stream_ptr()),
                   'tnp_collision_loss_backward')
        return d_pred, None, None, None


def CollisionLoss(predictions, batch_split, col_wt=10.0, col_distance=0.2):
    """Penalises primary predictions that come closer than col_distance to a neighbour (lstm/loss.py:138-162).
    Differentiable with respect to the primaries' positions, like the reference's."""
    _lib.require_device(predictions, 'predictions')
    if predictions.requires_grad and torch.is_grad_enabled():
        return _CollisionLossFn.apply(predictions, batch_split, col_wt, col_distance)
    pred = _lib.f32c(predictions.detach())
    return _collision_forward(pred, _lib.SceneIndex.get(batch_split, pred.device), col_wt, col_distance)


class PredictionLoss(torch.nn.Module):
    """2D Gaussian with a flat background (lstm/loss.py:6-91):  p(x) = 0.2 * N(x|mu, 3.0) + 0.8 * N(x|mu, sigma)."""

    def __init__(self, keep_batch_dim=False, background_rate=0.2, col_wt=0.0, col_distance=0.2):
        super(PredictionLoss, self).__init__()
        self.keep_batch_dim = keep_batch_dim
        self.background_rate = background_rate
        self.loss_multiplier = 1
        self.col_wt = col_wt
        self.col_distance = col_distance

    def terms(self, inputs, targets, batch_split, positions=None, tail=None, times=1.0):
        """(mean term, sum term or None): the primaries' NLL is a mean over frames x scenes (:85-91), the collision penalty
        a sum over scenes added un-normalised (:90) -- scene-sharded training scales the two differently
        (lstm/train_step.batch_loss).  ``tail`` / ``times`` (train_step only): the loss of inputs[-tail:], multiplied by
        ``times`` -- the trainer's ``criterion(rel_outputs[-pred_length:], ...) * batch_size`` (lstm/trainer.py:262-264)
        without the slice and multiplication nodes in the autograd graph."""
        loss = _primary_loss(0, inputs, targets, batch_split, self.background_rate, self.keep_batch_dim,
                             self.loss_multiplier * times, tail)
        if self.keep_batch_dim or not self.col_wt:
            return loss, None
        assert positions is not None, "Prediction positions required to calculate collision loss"
        return loss, CollisionLoss(positions, batch_split, self.col_wt, self.col_distance) * (self.loss_multiplier * times)

    def forward(self, inputs, targets, batch_split, positions=None):
        loss, col = self.terms(inputs, targets, batch_split, positions)
        return loss if col is None else loss + col


class L2Loss(torch.nn.Module):
    """L2 loss on the primaries, x100 (lstm/loss.py:93-135)."""

    def __init__(self, keep_batch_dim=False, col_wt=0.0, col_distance=0.2):
        super(L2Loss, self).__init__()
        self.keep_batch_dim = keep_batch_dim
        self.loss_multiplier = 100
        self.col_wt = col_wt
        self.col_distance = col_distance

    def terms(self, inputs, targets, batch_split, positions=None, tail=None, times=1.0):
        """(mean term, sum term or None), see PredictionLoss.terms"""
        # MSE over (t, scene, 2 coordinates): the kernel sums the two squared errors, hence the 1/2
        loss = _primary_loss(1, inputs, targets, batch_split, 0.0, self.keep_batch_dim, 0.5 * self.loss_multiplier * times, tail)
        if self.keep_batch_dim or not self.col_wt:
            return loss, None
        assert positions is not None, "Prediction positions required to calculate collision loss"
        return loss, CollisionLoss(positions, batch_split, self.col_wt, self.col_distance) * (self.loss_multiplier * times)

    def forward(self, inputs, targets, batch_split, positions=None):
        loss, col = self.terms(inputs, targets, batch_split, positions)
        return loss if col is None else loss + col


# ---- S-GAN trainer losses (reference lstm/loss.py:165-208, sgan/trainer.py:371-400) ---------------------------
def bce_loss(input_, target):
    """Numerically stable binary cross-entropy on logits, mean over the batch (reference lstm/loss.py:165-181)."""
    neg_abs = -input_.abs()
    loss = input_.clamp(min=0) - input_ * target + (1 + neg_abs.exp()).log()
    return loss.mean()


def gan_g_loss(scores_fake):
    """Generator loss with the reference's noisy label drawn from Python's `random` (lstm/loss.py:183-192)."""
    import random
    y_fake = torch.ones_like(scores_fake) * random.uniform(0.7, 1.2)
    return bce_loss(scores_fake, y_fake)


def gan_d_loss(scores_real, scores_fake):
    """Discriminator loss; the fake label is zeros * uniform = 0, the draw is kept for RNG-stream parity
    (lstm/loss.py:195-208)."""
    import random
    y_real = torch.ones_like(scores_real) * random.uniform(0.7, 1.2)
    y_fake = torch.zeros_like(scores_fake) * random.uniform(0, 0.3)
    return bce_loss(scores_real, y_real) + bce_loss(scores_fake, y_fake)


def variety_loss(criterion, inputs, target, batch_split, pred_length=12):
    """Top-k ("variety") loss of S-GAN: per scene the minimum over the k samples of the per-scene loss, summed over
    scenes (reference sgan/trainer.py:371-400).  `criterion` must keep the batch dimension."""
    per_sample = torch.stack([criterion(sample[-pred_length:], target, batch_split) for sample in inputs])
    return torch.min(per_sample, dim=0)[0].sum()
