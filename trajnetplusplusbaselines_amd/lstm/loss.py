"""Losses of the reference's lstm/loss.py (same class names and call signatures), evaluated on the primaries of a
batch by csrc/loss.hip.  Forward values only in this round (the backward arrives with the training kernels)."""
import torch

from .. import _lib


def _primary_loss(mode, inputs, targets, batch_split, background_rate, keep_batch_dim, scale):
    _lib.require_device(inputs, 'inputs')
    dev = inputs.device
    inputs = _lib.f32c(inputs.detach())
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


def CollisionLoss(predictions, batch_split, col_wt=10.0, col_distance=0.2):
    """Penalises primary predictions that come closer than col_distance to a neighbour (lstm/loss.py:138-162)."""
    _lib.require_device(predictions, 'predictions')
    dev = predictions.device
    pred = _lib.f32c(predictions.detach())
    T, M, ld = pred.shape
    idx = _lib.SceneIndex.get(batch_split, dev)
    partial = torch.empty(idx.B, dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().tnp_collision_loss_forward(_lib.ptr(pred), ld, _lib.ptr(idx.starts), idx.B, T, M, float(col_wt),
                                                     float(col_distance), _lib.ptr(partial), _lib.ptr(out),
                                                     _lib.stream_ptr()), 'tnp_collision_loss_forward')
    return out[0]


class PredictionLoss(torch.nn.Module):
    """2D Gaussian with a flat background (lstm/loss.py:6-91):  p(x) = 0.2 * N(x|mu, 3.0) + 0.8 * N(x|mu, sigma)."""

    def __init__(self, keep_batch_dim=False, background_rate=0.2, col_wt=0.0, col_distance=0.2):
        super(PredictionLoss, self).__init__()
        self.keep_batch_dim = keep_batch_dim
        self.background_rate = background_rate
        self.loss_multiplier = 1
        self.col_wt = col_wt
        self.col_distance = col_distance

    def forward(self, inputs, targets, batch_split, positions=None):
        loss = _primary_loss(0, inputs, targets, batch_split, self.background_rate, self.keep_batch_dim,
                             self.loss_multiplier)
        if self.keep_batch_dim:
            return loss
        if self.col_wt:
            assert positions is not None, "Prediction positions required to calculate collision loss"
            return loss + CollisionLoss(positions, batch_split, self.col_wt, self.col_distance) * self.loss_multiplier
        return loss


class L2Loss(torch.nn.Module):
    """L2 loss on the primaries, x100 (lstm/loss.py:93-135)."""

    def __init__(self, keep_batch_dim=False, col_wt=0.0, col_distance=0.2):
        super(L2Loss, self).__init__()
        self.keep_batch_dim = keep_batch_dim
        self.loss_multiplier = 100
        self.col_wt = col_wt
        self.col_distance = col_distance

    def forward(self, inputs, targets, batch_split, positions=None):
        # MSE over (t, scene, 2 coordinates): the kernel sums the two squared errors, hence the 1/2
        loss = _primary_loss(1, inputs, targets, batch_split, 0.0, self.keep_batch_dim, 0.5 * self.loss_multiplier)
        if self.keep_batch_dim:
            return loss
        if self.col_wt:
            assert positions is not None, "Prediction positions required to calculate collision loss"
            return loss + CollisionLoss(positions, batch_split, self.col_wt, self.col_distance) * self.loss_multiplier
        return loss
