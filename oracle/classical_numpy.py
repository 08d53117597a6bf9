"""Second, INDEPENDENT restatement of the classical predictors -- numpy only, TEST INFRASTRUCTURE ONLY.

``oracle/classical_oracle.c`` compiles the product's own ``csrc/classical_core.h`` with gcc, so it checks the GPU
*execution* (batching, LDS staging, cadence) but would share any formula bug with the kernels.  This module does not
include, import or translate that header: it restates the published algorithms a second time, in matrix / vectorised
form, so that a slip in one restatement shows up as a disagreement between the two
(``tests/test_classical.py::test_independent_*``).

PARITY STAYS UNPINNED for these rows: the reference wraps un-vendored third-party packages
(classical/socialforce.py:6-8,84-95 -> svenkreiss/socialforce; classical/orca.py:4,90-119 -> RVO2 via Python-RVO2;
classical/kalman.py:2,40-60 -> pykalman), none present in /root/reference, none installable here, and no reference test
touches classical/.  Two restatements that agree are still two restatements of what was recalled from those packages'
published algorithms:
  * Helbing & Molnar (1995) social force in the socialforce-0.1 form: relaxation to the desired velocity, elliptical
    pedestrian potential V = v0 exp(-b / sigma), gradient by forward differences (1e-3), 200 degree field of view with
    weight 0.5 outside, speed cap 1.3 x initial speed, explicit Euler at delta_t;
  * van den Berg et al. (2011) ORCA as in RVO2 2.0: k nearest neighbours inside neighborDist, one half-plane per
    neighbour, 2-D linear programme with the 3-D fallback, RVO_EPSILON = 1e-5, float32;
  * Shumway & Stoffer EM for a linear-Gaussian state-space model as pykalman runs it (transition / observation
    covariance, initial mean / covariance; RTS smoother), then the mean of sampled observation sequences.
"""
import numpy as np

# ------------------------------------------------------------------------------------------------------------
# social force (float64)
# ------------------------------------------------------------------------------------------------------------


def _sf_b(r, speeds, dirs, delta_t):
    """Semi-minor axis of the ellipse, r [n, n, 2] = r_a - r_b, speeds / dirs of pedestrian b."""
    step = (delta_t * speeds)[None, :, None] * dirs[None, :, :]
    n1 = np.linalg.norm(r, axis=-1)
    n2 = np.linalg.norm(r - step, axis=-1)
    with np.errstate(invalid='ignore'):                     # the diagonal (a pedestrian and itself) is masked by the caller
        return 0.5 * np.sqrt((n1 + n2) ** 2 - (delta_t * speeds)[None, :] ** 2)


def sf_step(pos, vel, goal, speed0, tau=0.5, v0=2.1, sigma=0.3, delta_t=0.05):
    """One explicit-Euler step of one scene: pos, vel, goal [n, 2], speed0 [n] (initial speeds, the cap is 1.3 x) ->
    (pos, vel) after the step.  ``sf_rollout`` and the driving stub of ``oracle/classical_stubs.py`` both run this."""
    n = pos.shape[0]
    cosphi = np.cos(np.deg2rad(100.0))
    eps = 1e-3
    e = goal - pos
    e = e / np.linalg.norm(e, axis=1, keepdims=True)
    force = (speed0[:, None] * e - vel) / tau
    if n > 1:
        r = pos[:, None, :] - pos[None, :, :]
        sp = np.linalg.norm(vel, axis=1)
        pot = lambda rr: v0 * np.exp(-_sf_b(rr, sp, e, delta_t) / sigma)
        v = pot(r)
        grad = np.stack([(pot(r + np.array([eps, 0.0])) - v) / eps, (pot(r + np.array([0.0, eps])) - v) / eps], axis=-1)
        f = -grad                                               # force on a from b
        seen = np.einsum('ak,abk->ab', e, -f) > np.linalg.norm(f, axis=-1) * cosphi
        w = np.where(seen, 1.0, 0.5)
        np.fill_diagonal(w, 0.0)
        f[np.arange(n), np.arange(n)] = 0.0
        force = force + (w[..., None] * f).sum(axis=1)
    wv = vel + delta_t * force
    ws = np.linalg.norm(wv, axis=1)
    vel = wv * np.minimum(1.0, 1.3 * speed0 / ws)[:, None]
    return pos + vel * delta_t, vel


def sf_rollout(state0, scene_start, n_steps=96, sample_every=8, tau=0.5, v0=2.1, sigma=0.3, delta_t=0.05):
    """state0 [M, 6] = x, y, vx, vy, goal_x, goal_y -> positions after steps 1, 1 + sample_every, ... [n_out, M, 2]
    (the wrapper keeps every 8th of the post-step states starting with the first, classical/socialforce.py:95)."""
    state0 = np.asarray(state0, dtype=np.float64)
    M = state0.shape[0]
    out = np.empty(((n_steps + sample_every - 1) // sample_every, M, 2))
    for lo, hi in zip(scene_start[:-1], scene_start[1:]):
        pos, vel, goal = state0[lo:hi, 0:2].copy(), state0[lo:hi, 2:4].copy(), state0[lo:hi, 4:6]
        speed0 = np.linalg.norm(vel, axis=1)
        k = 0
        for step in range(n_steps):
            pos, vel = sf_step(pos, vel, goal, speed0, tau, v0, sigma, delta_t)
            if step % sample_every == 0:
                out[k, lo:hi] = pos
                k += 1
    return out


# ------------------------------------------------------------------------------------------------------------
# constant-velocity Kalman filter with EM (float64, matrix form)
# ------------------------------------------------------------------------------------------------------------
_A = np.array([[1., 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1], [0, 0, 0, 1]])
_C = np.array([[1., 0, 0, 0], [0, 0, 1, 0]])


def _kf_smooth(obs, Q, R, m0, P0):
    T = obs.shape[0]
    xp, Pp, xf, Pf = np.zeros((T, 4)), np.zeros((T, 4, 4)), np.zeros((T, 4)), np.zeros((T, 4, 4))
    for t in range(T):
        if t == 0:
            xp[t], Pp[t] = m0, P0
        else:
            xp[t], Pp[t] = _A @ xf[t - 1], _A @ Pf[t - 1] @ _A.T + Q
        S = _C @ Pp[t] @ _C.T + R
        K = Pp[t] @ _C.T @ np.linalg.inv(S)
        xf[t] = xp[t] + K @ (obs[t] - _C @ xp[t])
        Pf[t] = Pp[t] - K @ _C @ Pp[t]
    xs, Ps, G = xf.copy(), Pf.copy(), np.zeros((T - 1, 4, 4))
    for t in range(T - 2, -1, -1):
        G[t] = Pf[t] @ _A.T @ np.linalg.inv(Pp[t + 1])
        xs[t] = xf[t] + G[t] @ (xs[t + 1] - xp[t + 1])
        Ps[t] = Pf[t] + G[t] @ (Ps[t + 1] - Pp[t + 1]) @ G[t].T
    return xs, Ps, G


def kalman_em(obs, n_iter=10, transition_var=1e-5, observation_var=0.05 ** 2):
    """EM over transition covariance, observation covariance, initial mean and initial covariance (pykalman's default
    ``em_vars``) for one track obs [T, 2]; initial mean (x0, 0, y0, 0), initial covariance I (classical/kalman.py:32-49).
    -> Q, R, m0, P0"""
    T = obs.shape[0]
    Q, R = transition_var * np.eye(4), observation_var * np.eye(2)
    m0, P0 = np.array([obs[0, 0], 0.0, obs[0, 1], 0.0]), np.eye(4)
    for _ in range(n_iter):
        xs, Ps, G = _kf_smooth(obs, Q, R, m0, P0)
        err = obs - xs @ _C.T
        R = (np.einsum('ti,tj->ij', err, err) + (_C @ Ps @ _C.T).sum(axis=0)) / T
        if T > 1:
            d = xs[1:] - xs[:-1] @ _A.T
            pair = Ps[1:] @ np.transpose(G, (0, 2, 1))            # Cov(x_{t+1}, x_t | all observations)
            Q = (np.einsum('ti,tj->ij', d, d) + (_A @ Ps[:-1] @ _A.T).sum(axis=0) + Ps[1:].sum(axis=0)
                 - (pair @ _A.T).sum(axis=0) - (_A @ np.transpose(pair, (0, 2, 1))).sum(axis=0)) / (T - 1)
        m0, P0 = xs[0], Ps[0]
    return Q, R, m0, P0


def kalman_sample(x0, Q, R, z):
    """One sampled continuation: z [n_steps, 6] standard normal draws (4 state + 2 observation components per step);
    step 0 is x0 itself (no transition noise), every step emits C x + observation noise.  -> (states, observations)"""
    LQ, LR = _chol_psd(Q), _chol_psd(R)
    n_steps = z.shape[0]
    xs, ys = np.zeros((n_steps, 4)), np.zeros((n_steps, 2))
    x = np.asarray(x0, dtype=np.float64).copy()
    for t in range(n_steps):
        if t > 0:
            x = _A @ x + LQ @ z[t, :4]
        xs[t] = x
        ys[t] = _C @ x + LR @ z[t, 4:6]
    return xs, ys


def kalman_predict(obs, z, n_iter=10, transition_var=1e-5, observation_var=0.05 ** 2):
    """obs [n, T, 2], z [n, n_samples, n_steps, 6] standard normal draws (4 state + 2 observation components per step)
    -> mean sampled observations [n, n_steps, 2]; step 0 is the last smoothed state itself (no transition noise)."""
    obs, z = np.asarray(obs, dtype=np.float64), np.asarray(z, dtype=np.float64)
    n = obs.shape[0]
    n_samples, n_steps = z.shape[1], z.shape[2]
    out = np.zeros((n, n_steps, 2))
    for i in range(n):
        Q, R, m0, P0 = kalman_em(obs[i], n_iter, transition_var, observation_var)
        xs, _, _ = _kf_smooth(obs[i], Q, R, m0, P0)
        for s in range(n_samples):
            out[i] += kalman_sample(xs[-1], Q, R, z[i, s])[1]
        out[i] /= n_samples
    return out


def _chol_psd(M):
    """lower Cholesky factor that tolerates a singular (PSD) matrix: zero pivots give zero columns"""
    n = M.shape[0]
    L = np.zeros_like(M)
    for j in range(n):
        s = M[j, j] - L[j, :j] @ L[j, :j]
        d = np.sqrt(s) if s > 0 else 0.0
        L[j, j] = d
        for i in range(j + 1, n):
            L[i, j] = (M[i, j] - L[i, :j] @ L[j, :j]) / d if d > 0 else 0.0
    return L


# ------------------------------------------------------------------------------------------------------------
# ORCA (float32): objects instead of flat loops -- a Line is (point, direction), velocities are 2-vectors
# ------------------------------------------------------------------------------------------------------------
_F = np.float32
_EPS = _F(1e-5)


def _det(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _norm(v):
    return v * (_F(1.0) / np.sqrt(v @ v))


def _lp_line(lines, k, radius, opt, direction):
    """optimum on line k inside the disc and the half-planes 0..k-1, or None"""
    p, d = lines[k]
    dot = p @ d
    disc = dot * dot + radius * radius - p @ p
    if disc < 0:
        return None
    root = np.sqrt(disc)
    lo, hi = -dot - root, -dot + root
    for q, e in lines[:k]:
        den, num = _det(d, e), _det(e, p - q)
        if abs(den) <= _EPS:
            if num < 0:
                return None
            continue
        t = num / den
        if den >= 0:
            hi = min(hi, t)
        else:
            lo = max(lo, t)
        if lo > hi:
            return None
    if direction:
        t = hi if opt @ d > 0 else lo
    else:
        t = min(max(d @ (opt - p), lo), hi)
    return p + t * d


def _lp_plane(lines, radius, opt, direction):
    if direction:
        res = opt * radius
    elif opt @ opt > radius * radius:
        res = _norm(opt) * radius
    else:
        res = opt.copy()
    for k, (p, d) in enumerate(lines):
        if _det(d, p - res) > 0:
            new = _lp_line(lines, k, radius, opt, direction)
            if new is None:
                return k, res
            res = new
    return len(lines), res


def _lp_fallback(lines, start, radius, res):
    dist = _F(0.0)
    for i in range(start, len(lines)):
        p, d = lines[i]
        if _det(d, p - res) > dist:
            proj = []
            for q, e in lines[:i]:
                dt = _det(d, e)
                if abs(dt) <= _EPS:
                    if d @ e > 0:
                        continue
                    point = _F(0.5) * (p + q)
                else:
                    point = p + (_det(e, p - q) / dt) * d
                proj.append((point, _norm(e - d)))
            fail, new = _lp_plane(proj, radius, np.array([-d[1], d[0]], dtype=_F), True)
            if fail == len(proj):
                res = new
            dist = _det(d, p - res)
    return res


def orca_new_velocity(a, pos, vel, pref, max_speed, time_step, neighbor_dist, max_neighbors, time_horizon, radius):
    """RVO2 Agent::computeNeighbors + computeNewVelocity for agent a of one scene (all float32); returns (v, neighbours)"""
    n = pos.shape[0]
    rng_sq = neighbor_dist * neighbor_dist
    nbrs = []                                                          # (dist_sq, index), ascending, at most max_neighbors
    for b in range(n):
        if b == a:
            continue
        dv = pos[a] - pos[b]
        dsq = dv @ dv
        if dsq < rng_sq:
            if len(nbrs) < max_neighbors:
                nbrs.append((dsq, b))
            else:
                nbrs[-1] = (dsq, b)
            i = len(nbrs) - 1
            while i > 0 and nbrs[i][0] < nbrs[i - 1][0]:
                nbrs[i], nbrs[i - 1] = nbrs[i - 1], nbrs[i]
                i -= 1
            if len(nbrs) == max_neighbors:
                rng_sq = nbrs[-1][0]
    inv_h = _F(1.0) / time_horizon
    lines = []
    for _, b in nbrs:
        rp, rv = pos[b] - pos[a], vel[a] - vel[b]
        dsq, R = rp @ rp, radius + radius
        if dsq > R * R:
            w = rv - inv_h * rp
            wsq, dot = w @ w, w @ rp
            if dot < 0 and dot * dot > R * R * wsq:                    # project on the cut-off circle
                wl = np.sqrt(wsq)
                uw = w * (_F(1.0) / wl)
                d = np.array([uw[1], -uw[0]], dtype=_F)
                u = (R * inv_h - wl) * uw
            else:                                                      # project on a leg
                leg = np.sqrt(dsq - R * R)
                if _det(rp, w) > 0:
                    d = np.array([rp[0] * leg - rp[1] * R, rp[0] * R + rp[1] * leg], dtype=_F) * (_F(1.0) / dsq)
                else:
                    d = -np.array([rp[0] * leg + rp[1] * R, -rp[0] * R + rp[1] * leg], dtype=_F) * (_F(1.0) / dsq)
                u = (rv @ d) * d - rv
        else:                                                          # already colliding: resolve within one time step
            inv_t = _F(1.0) / time_step
            w = rv - inv_t * rp
            wl = np.sqrt(w @ w)
            uw = w * (_F(1.0) / wl)
            d = np.array([uw[1], -uw[0]], dtype=_F)
            u = (R * inv_t - wl) * uw
        lines.append((vel[a] + _F(0.5) * u, d))
    fail, res = _lp_plane(lines, max_speed, pref, False)
    if fail < len(lines):
        res = _lp_fallback(lines, fail, max_speed, res)
    return res, [b for _, b in nbrs]


def orca_rollout(pos0, vel0, goals, speed, max_speed, scene_start, n_iter=97, sample_every=8, time_step=0.05,
                 neighbor_dist=1.5, max_neighbors=10, time_horizon=1.5, radius=0.4, want_neighbors=False):
    """rvo2 doStep() x n_iter with the wrapper's preferred-velocity rule (classical/orca.py:99-119: positions sampled when
    count % 8 == 0, preferred velocity = goal vector capped to the initial speed, zero within 0.05 m; float64 there)."""
    pos0, vel0 = np.asarray(pos0, dtype=_F), np.asarray(vel0, dtype=_F)
    goals, speed = np.asarray(goals, dtype=np.float64), np.asarray(speed, dtype=np.float64)
    max_speed = np.asarray(max_speed, dtype=_F)
    M = pos0.shape[0]
    out = np.empty((n_iter // sample_every, M, 2), dtype=_F)
    first = np.full((M, 16), -1, dtype=np.int32)
    ts, nd, th, rad = _F(time_step), _F(neighbor_dist), _F(time_horizon), _F(radius)
    for lo, hi in zip(scene_start[:-1], scene_start[1:]):
        pos, vel = pos0[lo:hi].copy(), vel0[lo:hi].copy()
        pref = np.zeros_like(pos)
        k = 0
        for count in range(1, n_iter + 1):
            new = np.empty_like(vel)
            for a in range(hi - lo):
                new[a], nb = orca_new_velocity(a, pos, vel, pref[a], max_speed[lo + a], ts, nd, max_neighbors, th, rad)
                if count == 1:
                    first[lo + a, :len(nb)] = nb
            vel = new
            pos = pos + vel * ts
            if count % sample_every == 0:
                out[k, lo:hi] = pos
                k += 1
            to_goal = goals[lo:hi] - pos.astype(np.float64)
            dist = np.linalg.norm(to_goal, axis=1)
            capped = np.where((dist > speed[lo:hi])[:, None], speed[lo:hi, None] * to_goal / np.where(dist > 0, dist, 1.0)[:, None], to_goal)
            pref = np.where((dist < 0.05)[:, None], 0.0, capped).astype(_F)
    return (out, first) if want_neighbors else out
