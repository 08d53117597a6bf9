"""Driving stubs for the reference's classical wrappers -- TEST INFRASTRUCTURE ONLY (build container only).

The reference's ``classical/socialforce.py``, ``classical/orca.py`` and ``classical/kalman.py`` are thin wrappers around
three third-party packages that are NOT in /root/reference and cannot be installed here (svenkreiss/socialforce,
Python-RVO2, pykalman).  What the wrappers themselves decide IS reference code: which agents take part, the stride-3
initial velocity, the extrapolated goal, max_speed = 1.3 x speed, the 96-step / every-8th-state cadence of the social
force rollout, ORCA's 97 ``doStep`` calls sampled at count 8..96 with the per-step preferred-velocity rule, Kalman's
matrices and the 13-sample / mean-of-5 rule.  To pin THAT half to the reference's own lines, ``install()`` registers
modules named ``socialforce``, ``rvo2`` and ``pykalman`` in ``sys.modules`` that expose exactly the calls the wrappers
make and delegate the simulator arithmetic to ``oracle/classical_numpy.py``; the unmodified reference ``predict()``
functions then run on top of them (``oracle/gen_golden_r5.py``) and every call they make is recorded.

The third-party ARITHMETIC stays unpinned (it is our restatement behind the stub); the stubs' call surface follows the
call sites in the reference:
  socialforce.Simulator(initial_state, ped_ped=, field_of_view=, delta_t=, tau=), .step() -> self, .state
      (classical/socialforce.py:86-91); socialforce.potentials.PedPedPotential(delta_t, v0=, sigma=) (:86);
      socialforce.field_of_view.FieldOfView() (:87)
  rvo2.PyRVOSimulator(timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed)
      (classical/orca.py:91), .addAgent(pos, maxSpeed=, velocity=) (:55), .doStep() (:102), .getAgentPosition(i) (:106),
      .setAgentPrefVelocity(i, v) (:113, :119)
  pykalman.KalmanFilter(transition_matrices=, observation_matrices=, transition_covariance=, observation_covariance=,
      initial_state_mean=) (classical/kalman.py:40-44), .em(X) (:47), .smooth(X) (:48),
      .sample(n, initial_state=) (:55); the normal draws come from numpy's global RandomState as pykalman's do
      (``random_state=None``), so ``np.random.seed`` makes the run reproducible: each ``sample`` call consumes
      ``standard_normal((n, 6))`` (4 state + 2 observation components per step).
"""
import sys
import types

import numpy as np

from . import classical_numpy as cn

RECORD = {'sf': [], 'orca': [], 'kalman': []}
# 'numpy': the simulator arithmetic is oracle/classical_numpy.py (independent of the product);
# 'core':  it is the product's csrc/classical_core.h compiled for the host (oracle/classical_oracle.c, orc_*_step), the
#          arithmetic the HIP kernels execute -- so GPU == fixture can be demanded bit for bit (ORCA, float32) while every
#          decision between two simulator calls is still taken by the reference's own Python.
BACKEND = ['numpy']


def set_backend(name):
    assert name in ('numpy', 'core')
    BACKEND[0] = name


def _core():
    from . import oracle
    return oracle.lib()


def _pd(a):
    import ctypes
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _pf(a):
    import ctypes
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def reset_record():
    for v in RECORD.values():
        del v[:]


# ---------------------------------------------------------------------------------------------------------------
class PedPedPotential(object):
    def __init__(self, delta_t, v0=2.1, sigma=0.3):
        self.delta_t, self.v0, self.sigma = delta_t, v0, sigma


class FieldOfView(object):
    def __init__(self, twophi=200.0, out_of_view_factor=0.5):
        assert twophi == 200.0 and out_of_view_factor == 0.5


class Simulator(object):
    """state [n, 7]: x, y, vx, vy, goal_x, goal_y, tau (socialforce-0.1 appends tau as a seventh column)."""

    def __init__(self, initial_state, ped_ped=None, field_of_view=None, delta_t=0.4, tau=0.5):
        initial_state = np.asarray(initial_state, dtype=np.float64)
        assert initial_state.ndim == 2 and initial_state.shape[1] == 6
        self.delta_t, self.tau, self.ped_ped = delta_t, tau, ped_ped
        assert ped_ped is not None and field_of_view is not None and ped_ped.delta_t == delta_t
        self.state = np.concatenate([initial_state, np.full((initial_state.shape[0], 1), tau)], axis=1)
        self.speed0 = np.linalg.norm(initial_state[:, 2:4], axis=1)
        self.n_steps = 0
        RECORD['sf'].append({'initial_state': initial_state.copy(), 'delta_t': delta_t, 'tau': tau, 'v0': ped_ped.v0,
                             'sigma': ped_ped.sigma, 'sim': self})

    def step(self):
        if BACKEND[0] == 'core':
            import ctypes
            st = np.ascontiguousarray(self.state, dtype=np.float64).copy()
            isp = np.ascontiguousarray(self.speed0, dtype=np.float64)
            assert np.all(st[:, 6] == self.tau)
            _core().orc_sf_step(_pd(st), _pd(isp), st.shape[0], ctypes.c_double(self.ped_ped.v0),
                                ctypes.c_double(self.ped_ped.sigma), ctypes.c_double(self.delta_t))
            self.state = st
            self.n_steps += 1
            return self
        pos, vel = cn.sf_step(self.state[:, 0:2], self.state[:, 2:4], self.state[:, 4:6], self.speed0, self.tau,
                              self.ped_ped.v0, self.ped_ped.sigma, self.delta_t)
        self.state = self.state.copy()
        self.state[:, 0:2], self.state[:, 2:4] = pos, vel
        self.n_steps += 1
        return self


# ---------------------------------------------------------------------------------------------------------------
_F = np.float32


class PyRVOSimulator(object):
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed,
                 velocity=(0, 0)):
        self.time_step, self.neighbor_dist, self.max_neighbors = _F(timeStep), _F(neighborDist), int(maxNeighbors)
        self.time_horizon, self.radius, self.default_max_speed = _F(timeHorizon), _F(radius), _F(maxSpeed)
        self.pos, self.vel, self.pref, self.max_speed = [], [], [], []
        self.n_steps = 0
        self.rec = {'args': (timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed),
                    'agents': [], 'pref_calls': [], 'sim': self}
        RECORD['orca'].append(self.rec)

    def addAgent(self, pos, neighborDist=None, maxNeighbors=None, timeHorizon=None, timeHorizonObst=None, radius=None,
                 maxSpeed=None, velocity=None):
        assert neighborDist is None and maxNeighbors is None and timeHorizon is None and radius is None
        self.pos.append(np.array(pos, dtype=_F))
        self.vel.append(np.array(velocity if velocity is not None else (0, 0), dtype=_F))
        self.pref.append(np.zeros(2, dtype=_F))
        self.max_speed.append(_F(maxSpeed if maxSpeed is not None else self.default_max_speed))
        self.rec['agents'].append((tuple(pos), maxSpeed, tuple(velocity)))
        return len(self.pos) - 1

    def doStep(self):
        pos, vel = np.array(self.pos, dtype=_F).reshape(-1, 2), np.array(self.vel, dtype=_F).reshape(-1, 2)
        if BACKEND[0] == 'core':
            import ctypes
            prf = np.array(self.pref, dtype=_F).reshape(-1, 2)
            ms = np.array(self.max_speed, dtype=_F)
            _core().orc_orca_step(_pf(pos), _pf(vel), _pf(prf), _pf(ms), pos.shape[0], ctypes.c_float(self.time_step),
                                  ctypes.c_float(self.neighbor_dist), self.max_neighbors, ctypes.c_float(self.time_horizon),
                                  ctypes.c_float(self.radius))
            self.pos, self.vel = [p for p in pos], [v for v in vel]
            self.n_steps += 1
            return
        new = np.empty_like(vel)
        for a in range(pos.shape[0]):
            new[a], _ = cn.orca_new_velocity(a, pos, vel, self.pref[a], self.max_speed[a], self.time_step,
                                             self.neighbor_dist, self.max_neighbors, self.time_horizon, self.radius)
        pos = pos + new * self.time_step
        self.pos, self.vel = [p for p in pos], [v for v in new]
        self.n_steps += 1

    def getAgentPosition(self, i):
        return (float(self.pos[i][0]), float(self.pos[i][1]))

    def setAgentPrefVelocity(self, i, v):
        assert isinstance(v, tuple) and len(v) == 2
        self.pref[i] = np.array(v, dtype=_F)
        if len(self.rec['pref_calls']) < 4096:
            self.rec['pref_calls'].append((self.n_steps, i, float(v[0]), float(v[1])))


# ---------------------------------------------------------------------------------------------------------------
class KalmanFilter(object):
    def __init__(self, transition_matrices=None, observation_matrices=None, transition_covariance=None,
                 observation_covariance=None, initial_state_mean=None):
        assert np.array_equal(np.asarray(transition_matrices, dtype=np.float64), cn._A)
        assert np.array_equal(np.asarray(observation_matrices, dtype=np.float64), cn._C)
        tq, ro = np.asarray(transition_covariance), np.asarray(observation_covariance)
        assert np.array_equal(tq, tq[0, 0] * np.eye(4)) and np.array_equal(ro, ro[0, 0] * np.eye(2))
        self.transition_var, self.observation_var = float(tq[0, 0]), float(ro[0, 0])
        self.initial_state_mean = np.asarray(initial_state_mean, dtype=np.float64)
        self.params = None
        self.rec = {'transition_var': self.transition_var, 'observation_var': self.observation_var,
                    'initial_state_mean': self.initial_state_mean.copy(), 'sample_calls': []}
        RECORD['kalman'].append(self.rec)

    def em(self, X, n_iter=10):
        X = np.asarray(X, dtype=np.float64)
        # the wrapper's initial mean is (x0, 0, y0, 0) of the same track (classical/kalman.py:32)
        assert np.array_equal(self.initial_state_mean, [X[0, 0], 0.0, X[0, 1], 0.0])
        self.rec['em_obs'] = X.copy()
        if BACKEND[0] == 'core':
            import ctypes
            X = np.ascontiguousarray(X)
            self.model, self.x_last = np.zeros(46), np.zeros(4)
            _core().orc_kalman_em(_pd(X), X.shape[0], n_iter, ctypes.c_double(self.transition_var),
                                  ctypes.c_double(self.observation_var), _pd(self.model), _pd(self.x_last))
            return self
        self.params = cn.kalman_em(X, n_iter, self.transition_var, self.observation_var)
        return self

    def smooth(self, X):
        if BACKEND[0] == 'core':          # the wrapper only reads the last smoothed state (classical/kalman.py:48,55)
            xs = np.full((len(X), 4), np.nan)
            xs[-1] = self.x_last
            return xs, None
        Q, R, m0, P0 = self.params
        xs, Ps, _ = cn._kf_smooth(np.asarray(X, dtype=np.float64), Q, R, m0, P0)
        return xs, Ps

    def sample(self, n_timesteps, initial_state=None, random_state=None):
        assert initial_state is not None and random_state is None
        z = np.random.standard_normal((n_timesteps, 6))
        self.rec['sample_calls'].append((n_timesteps, z.copy()))
        if BACKEND[0] == 'core':
            out = np.zeros((n_timesteps, 2))
            x0 = np.ascontiguousarray(initial_state, dtype=np.float64)
            _core().orc_kalman_sample(_pd(self.model), _pd(x0), n_timesteps, _pd(z), _pd(out))
            return None, out
        Q, R, _, _ = self.params
        return cn.kalman_sample(initial_state, Q, R, z)


# ---------------------------------------------------------------------------------------------------------------
class _Reader(object):
    """trajnetplusplustools.Reader.paths_to_xy as classical/constant_velocity.py:8 calls it: frames of the primary,
    float64 [T, N, 2], NaN for absent."""

    @staticmethod
    def paths_to_xy(paths):
        frames = sorted(set(r.frame for r in paths[0]))
        index = {f: i for i, f in enumerate(frames)}
        xy = np.full((len(frames), len(paths), 2), np.nan)
        for p, path in enumerate(paths):
            for r in path:
                if r.frame in index:
                    xy[index[r.frame], p] = [r.x, r.y]
        return xy


def install():
    """Register the driving stubs and return the reference's (socialforce, orca, kalman, constant_velocity) modules,
    imported FROM /root/reference/trajnetbaselines/classical/ by file so that the package __init__ (which pulls in the
    evaluators) is not needed."""
    import importlib.util
    import os
    from . import ref_import
    if not ref_import.available():
        raise RuntimeError('reference checkout not found')
    sf = types.ModuleType('socialforce')
    sf.Simulator = Simulator
    sf.potentials = types.ModuleType('socialforce.potentials')
    sf.potentials.PedPedPotential = PedPedPotential
    sf.field_of_view = types.ModuleType('socialforce.field_of_view')
    sf.field_of_view.FieldOfView = FieldOfView
    rvo2 = types.ModuleType('rvo2')
    rvo2.PyRVOSimulator = PyRVOSimulator
    pk = types.ModuleType('pykalman')
    pk.KalmanFilter = KalmanFilter
    tools = sys.modules.get('trajnetplusplustools') or types.ModuleType('trajnetplusplustools')
    tools.Reader = _Reader
    saved = {}
    for name, mod in [('socialforce', sf), ('socialforce.potentials', sf.potentials),
                      ('socialforce.field_of_view', sf.field_of_view), ('rvo2', rvo2), ('pykalman', pk),
                      ('trajnetplusplustools', tools)]:
        saved[name] = sys.modules.get(name)
        sys.modules[name] = mod
    out = []
    root = os.path.join(ref_import.REFERENCE_ROOT, 'trajnetbaselines', 'classical')
    for name in ['socialforce', 'orca', 'kalman', 'constant_velocity']:
        spec = importlib.util.spec_from_file_location('_ref_classical_' + name, os.path.join(root, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out.append(mod)
    # leave sys.modules as found: other tests import the reference with EMPTY stubs (ref_import.import_reference)
    for name, mod in saved.items():
        if mod is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = mod
    return tuple(out)
