"""Synthetic crowds of the shape BASELINE.md section 3 / SURVEY.md 8(d) prescribe.

``xy[t, m] = x0[m] + v[m] * t + eps``,  t = 0..T-1,  x0 ~ U(-4, 4)^2,
v ~ N(0, 0.3^2) m/frame, eps ~ N(0, 0.02^2); ``torch.Generator().manual_seed(seed)``.
Regime (ii): ragged scenes and tracks that enter / leave (NaN before a start
frame / after an end frame), as in the real TrajNet++ data.
"""
import torch


def linear_crowd(scenes, agents, frames=21, seed=0, dtype=torch.float32):
    """All agents present. Returns (xy [frames, scenes*agents, 2], batch_split [scenes+1] int64)."""
    g = torch.Generator().manual_seed(seed)
    m = scenes * agents
    x0 = torch.rand(m, 2, generator=g) * 8.0 - 4.0
    v = torch.randn(m, 2, generator=g) * 0.3
    t = torch.arange(frames, dtype=torch.float32).view(frames, 1, 1)
    eps = torch.randn(frames, m, 2, generator=g) * 0.02
    xy = (x0.unsqueeze(0) + v.unsqueeze(0) * t + eps).to(dtype)
    split = torch.arange(0, m + 1, agents, dtype=torch.int64)
    return xy, split


def ragged_crowd(scenes, min_agents=2, max_agents=12, frames=21, obs_length=9, seed=0, nan_frac=0.2):
    """Ragged scenes with entering / leaving neighbours (primary = row batch_split[s], always present).

    Returns (xy [frames, M, 2] with NaN for absent entries, batch_split [scenes+1] int64).
    """
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(min_agents, max_agents + 1, (scenes,), generator=g)
    split = torch.zeros(scenes + 1, dtype=torch.int64)
    split[1:] = torch.cumsum(sizes, 0)
    m = int(split[-1])
    x0 = torch.rand(m, 2, generator=g) * 8.0 - 4.0
    v = torch.randn(m, 2, generator=g) * 0.3
    t = torch.arange(frames, dtype=torch.float32).view(frames, 1, 1)
    eps = torch.randn(frames, m, 2, generator=g) * 0.02
    xy = x0.unsqueeze(0) + v.unsqueeze(0) * t + eps
    # presence windows for non-primary tracks
    u = torch.rand(m, generator=g)
    lo = torch.randint(0, frames, (m,), generator=g)
    hi = torch.randint(0, frames, (m,), generator=g)
    start = torch.minimum(lo, hi)
    end = torch.maximum(lo, hi)
    partial = u < min(1.0, 2.5 * nan_frac)
    partial[split[:-1]] = False
    frame_idx = torch.arange(frames).view(frames, 1)
    absent = partial.view(1, m) & ((frame_idx < start.view(1, m)) | (frame_idx > end.view(1, m)))
    xy[absent.unsqueeze(-1).expand(-1, -1, 2)] = float('nan')
    return xy, split
