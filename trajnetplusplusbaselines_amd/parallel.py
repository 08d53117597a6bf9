"""Scene-level data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

Scenes are independent (all interaction is intra-scene, reference lstm/lstm.py:36-40), so inference shards scenes
across ranks with NO collective on the data path; ranks only meet to gather predictions (optional) or to reduce
timing / metrics.  Training adds exactly one gradient all-reduce per optimizer step; the
loss-scaling rule that reproduces the single-process gradient is `scale_loss_for_sharding`.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch_split, rank, world_size, balance='pairs'):
    """Contiguous block of scenes for `rank`.  Returns (scene_lo, scene_hi).

    balance = 'scenes': equal scene counts; 'tracks': equal sum of tracks; 'pairs': equal sum of N^2
    (the O(N^2) neighbour work), which is what matters for ragged batches."""
    split = torch.as_tensor(batch_split, dtype=torch.int64).cpu()
    n_scenes = split.numel() - 1
    if world_size <= 1:
        return 0, n_scenes
    sizes = (split[1:] - split[:-1]).to(torch.float64)
    if balance == 'scenes':
        w = torch.ones_like(sizes)
    elif balance == 'tracks':
        w = sizes
    else:
        w = sizes * sizes
    cum = torch.cumsum(w, 0)
    total = float(cum[-1]) if n_scenes else 0.0
    # boundary r = first scene index whose cumulative weight exceeds r/world of the total
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        idx = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64), right=False))
        bounds.append(max(bounds[-1], min(idx + 1 if n_scenes and float(cum[min(idx, n_scenes - 1)]) <= target else idx, n_scenes)))
    bounds.append(n_scenes)
    return bounds[rank], bounds[rank + 1]


def shard_batch(observed, goals, batch_split, rank, world_size, prediction_truth=None, balance='pairs'):
    """Slice one batch (track-major tensors [T, M, 2] / [M, 2], batch_split [B+1]) to this rank's scenes and
    rebase its batch_split.  Returns (observed, goals, batch_split, prediction_truth, (track_lo, track_hi))."""
    split = torch.as_tensor(batch_split, dtype=torch.int64).cpu()
    lo, hi = shard_bounds(split, rank, world_size, balance)
    t_lo, t_hi = int(split[lo]), int(split[hi])
    local_split = split[lo:hi + 1] - split[lo]
    obs = observed[:, t_lo:t_hi]
    g = goals[t_lo:t_hi] if goals is not None else None
    truth = prediction_truth[:, t_lo:t_hi] if prediction_truth is not None else None
    return obs, g, local_split, truth, (t_lo, t_hi)


def gather_tracks(local, n_tracks_total, track_range, group=None):
    """All-gather per-rank outputs [S, M_local, D] back into [S, M_total, D] (every rank gets the full tensor).
    Ranks may hold different numbers of tracks; shapes are exchanged first."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    ranges = [None] * world
    dist.all_gather_object(ranges, tuple(track_range), group=group)
    out = local.new_empty((local.shape[0], n_tracks_total) + tuple(local.shape[2:]))
    # all_gather wants equal shapes: pad every shard to the longest one
    longest = max(r[1] - r[0] for r in ranges)
    padded = local.new_zeros((local.shape[0], longest) + tuple(local.shape[2:]))
    padded[:, :local.shape[1]] = local
    pieces = [torch.empty_like(padded) for _ in ranges]
    dist.all_gather(pieces, padded, group=group)
    for r, p in zip(ranges, pieces):
        out[:, r[0]:r[1]] = p[:, :r[1] - r[0]]
    return out


def scale_loss_for_sharding(loss_mean_local, nominal_batch_size, n_local_scenes, n_global_scenes):
    """The reference multiplies the per-element mean loss by the nominal batch size (lstm/trainer.py:263).
    With scenes sharded over ranks and gradients SUM-reduced, `mean_local * batch_size * n_local / n_global`
    reproduces the single-process gradient exactly (mean over all primaries x batch_size)."""
    return loss_mean_local * nominal_batch_size * (float(n_local_scenes) / float(n_global_scenes))


def allreduce_gradients(parameters, group=None, bucket_bytes=32 << 20):
    """One flattened SUM all-reduce per bucket (parameters with grad None -- e.g. the unused goal embedding,
    lstm/lstm.py:75 -- are skipped consistently on every rank).  Social-LSTM is 19.7 MB of fp32 gradients:
    a single bucket, latency-bound on xGMI, so fewer larger messages is the right shape."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
    n_msgs = 0
    bucket, size = [], 0
    def flush():
        nonlocal bucket, size, n_msgs
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_msgs += 1
        bucket, size = [], 0
    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return n_msgs


def max_over_ranks(value, device=None, group=None):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
