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
    (the O(N^2) neighbour work), which is what matters for ragged batches.  A scene that straddles a target goes to
    the earlier rank, and every rank gets at least one scene whenever there are at least `world_size` scenes (an
    empty shard would leave its rank without a forward while the others wait in the all-reduce)."""
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
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first boundary whose prefix weight reaches the target: the straddling scene stays with the earlier rank
        b = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64), right=False)) + 1 if n_scenes else 0
        if n_scenes >= world_size:
            b = max(b, bounds[-1] + 1)                      # at least one scene for rank r-1 ...
            b = min(b, n_scenes - (world_size - r))         # ... and one left for every later rank
        bounds.append(max(bounds[-1], min(b, n_scenes)))
    bounds.append(n_scenes)
    return bounds[rank], bounds[rank + 1]


class Shard(tuple):
    """(observed, goals, batch_split, prediction_truth, track_range) of one rank -- unpacks like the 5-tuple it used to be --
    plus what the shard must know about the whole batch: ``pad_to`` (largest scene of the UNSHARDED batch: handed to
    ``LSTM.forward(..., pad_to=)`` it makes the shard compute exactly what the reference computes for the whole batch,
    whose scenes are all padded to that many slots, lstm/lstm.py:29), ``n_scenes_global`` and ``scene_range``."""

    def __new__(cls, observed, goals, batch_split, prediction_truth, track_range, pad_to, n_scenes_global, scene_range):
        self = super(Shard, cls).__new__(cls, (observed, goals, batch_split, prediction_truth, track_range))
        self.pad_to, self.n_scenes_global, self.scene_range = pad_to, n_scenes_global, scene_range
        return self

    observed = property(lambda self: self[0])
    goals = property(lambda self: self[1])
    batch_split = property(lambda self: self[2])
    prediction_truth = property(lambda self: self[3])
    track_range = property(lambda self: self[4])
    n_scenes = property(lambda self: self[2].numel() - 1)


def shard_batch(observed, goals, batch_split, rank, world_size, prediction_truth=None, balance='pairs'):
    """Slice one batch (track-major tensors [T, M, 2] / [M, 2], batch_split [B+1]) to this rank's scenes and
    rebase its batch_split.  Returns a ``Shard``: (observed, goals, batch_split, prediction_truth, (track_lo, track_hi))
    with ``.pad_to`` / ``.n_scenes_global`` / ``.scene_range``."""
    split = torch.as_tensor(batch_split, dtype=torch.int64).cpu()
    lo, hi = shard_bounds(split, rank, world_size, balance)
    t_lo, t_hi = int(split[lo]), int(split[hi])
    local_split = split[lo:hi + 1] - split[lo]
    obs = observed[:, t_lo:t_hi]
    g = goals[t_lo:t_hi] if goals is not None else None
    truth = prediction_truth[:, t_lo:t_hi] if prediction_truth is not None else None
    n_scenes = split.numel() - 1
    pad_to = int((split[1:] - split[:-1]).max()) if n_scenes > 0 else 0
    return Shard(obs, g, local_split, truth, (t_lo, t_hi), pad_to, n_scenes, (lo, hi))


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
    lstm/lstm.py:75 -- are skipped, consistently on every rank: ranks run the same model, and a rank whose shard is empty
    back-propagates a zero-weighted dummy scene, lstm/train_step.py).  Social-LSTM is 19.7 MB of fp32 gradients: a single
    bucket, latency-bound on xGMI, so fewer larger messages is the right shape.  Prefer ``GradBuckets`` on the training
    path: persistent flat buffers, no concatenation or copy-back, asynchronous launch."""
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


class GradBuckets(object):
    """Persistent flat gradient storage for data-parallel training: every trained parameter's ``.grad`` is a view into
    one of a few flat fp32 buffers (``bucket_bytes`` each, parameters in REVERSE registration order), so a bucket is
    all-reduced in place -- no ``torch.cat``, no copy-back -- with ``async_op=True``: on ROCm the collective runs on
    RCCL's own stream over xGMI while the compute stream carries on (the next bucket's launch, the optimizer's
    bookkeeping), and ``wait()`` only makes the compute stream wait for it (no host synchronisation).

    The buffers are laid out by ``attach()`` after the first backward: parameters that received no gradient (the goal
    embedding of a model without goals, lstm/lstm.py:75) stay out, keep ``grad is None`` and are skipped by the
    optimizer exactly as in the reference -- the same set on every rank, since ranks run the same model.  From then on
    gradients are zeroed in place (``zero()``), never set to None, so the views stay attached across steps."""

    def __init__(self, parameters, group=None, bucket_bytes=8 << 20):
        self.group = group
        self.params = [p for p in parameters if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.buckets = []       # (flat tensor, [params])
        self.attached = False
        self._pending = []

    def attach(self):
        cur, size = [], 0
        for p in reversed(self.params):
            if p.grad is None:
                continue
            cur.append(p)
            size += p.numel() * 4
            if size >= self.bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self.attached = True

    def _close(self, params):
        dev = params[0].device
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            off += p.numel()
        self.buckets.append((flat, list(params)))

    def zero(self):
        if not self.attached:
            for p in self.params:
                p.grad = None
            return
        for flat, _ in self.buckets:
            flat.zero_()

    def active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def launch_all(self):
        """Start the SUM all-reduce of every bucket (gradients enqueued on the current stream); returns the count."""
        if not self.attached:
            self.attach()
        if not self.active():
            return 0
        for flat, _ in self.buckets:
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return len(self.buckets)

    def wait(self):
        """The current stream waits for the launched collectives (no host block on ROCm / RCCL)."""
        for work in self._pending:
            work.wait()
        self._pending = []


class GradReducer(object):
    """Gradient all-reduce INSIDE the backward pass: ``LSTM``'s autograd function (lstm/training.py) calls this object
    with every gradient tensor the moment the kernel that produces it is enqueued -- largest first (the 16.8 MB first
    embedding layer of Social-LSTM) -- and gets back the asynchronous work handle of a SUM all-reduce.  On ROCm the
    collective runs on RCCL's stream over xGMI while the compute stream carries on with the remaining weight-gradient
    GEMMs (~0.9 ms at config 2, more than the ring all-reduce of 19.7 MB needs); the backward waits for the handles
    (stream-level wait, no host block) before it hands the gradients to autograd, so ``p.grad`` is already the sum over
    ranks when ``loss.backward()`` returns."""

    def __init__(self, group=None, coalesce_below=1 << 20):
        """``coalesce_below`` (bytes): gradients smaller than this are not sent one by one -- a ring all-reduce of a few KB
        is pure latency (tens of microseconds each, serialised on RCCL's stream: the thirteen small tensors of Social-LSTM
        would finish ~0.4 ms after the last weight-gradient GEMM) -- but collected and sent as ONE flattened message by
        ``flush()``, which the backward pass calls after its last gradient is enqueued.  0 = every tensor on its own."""
        self.group = group
        self.coalesce_below = int(coalesce_below)
        self.messages = 0
        self.bytes = 0
        self._small = []
        self._flushed = None          # handle of the last coalesced message until someone has waited for it

    def begin(self):
        """Start of a backward pass: drop gradients a previous pass collected but never sent (it raised before ``flush()``);
        they must not travel with this step's message."""
        self._small = []
        self._flushed = None

    def __call__(self, tensor):
        self.bytes += tensor.numel() * tensor.element_size()
        if tensor.numel() * tensor.element_size() < self.coalesce_below and tensor.dtype == torch.float32:
            self._small.append(tensor)
            return _Done(self, tensor)
        self.messages += 1
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def flush(self):
        """Send the collected small gradients as one flattened SUM all-reduce; returns a handle whose ``wait()`` makes the
        current stream wait for it and copies the sums back into the tensors (or None when nothing was collected)."""
        if not self._small:
            return None
        tensors, self._small = self._small, []
        flat = torch.cat([t.reshape(-1) for t in tensors])
        self.messages += 1
        self._flushed = _FlatWork(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat, tensors)
        return self._flushed


class _Done(object):
    """Handle of a gradient that travels with the coalesced message (GradReducer.flush).  The call-then-wait contract of a
    plain ``reduce_fn`` still holds: waiting on it before anyone called ``flush()`` sends the coalesced message now, and the
    wait covers that message -- a caller that never heard of ``flush()`` cannot end up with an unreduced gradient."""

    def __init__(self, reducer, tensor):
        self.reducer, self.tensor = reducer, tensor

    def wait(self):
        r = self.reducer
        if any(t is self.tensor for t in r._small):
            r.flush()
        w = r._flushed
        if w is not None and any(t is self.tensor for t in w.tensors):
            w.wait()
        return True


class _FlatWork(object):
    def __init__(self, work, flat, tensors):
        self.work, self.flat, self.tensors = work, flat, tensors

    def wait(self):
        if self.work is None:          # already waited for (a _Done handle and the backward's own flush handle may both wait)
            return True
        self.work.wait()
        self.work = None
        off = 0
        for t in self.tensors:
            t.copy_(self.flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        return True


def max_over_ranks(value, device=None, group=None):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
