"""Adam as the reference's trainers construct it (``torch.optim.Adam(model.parameters(), lr=..., weight_decay=...)``,
lstm/trainer.py:497, sgan/trainer.py:540-546) with the update executed by ONE native launch (csrc/optim.hip,
``tnp_adam_step``) instead of torch's nine per-operation multi-tensor kernels.

Same constructor arguments, ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) and semantics as
``torch.optim.Adam`` with ``amsgrad=False, maximize=False``: parameters whose ``.grad`` is None are skipped, weight decay
included; a checkpoint written by either optimiser loads into the other."""
import ctypes

import torch

from . import _lib


class AdamTensor(ctypes.Structure):
    """mirror of ``struct tnp_adam_tensor`` (include/trajnet_hip.h)"""
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p),
                ('exp_avg_sq', ctypes.c_void_p), ('n', ctypes.c_int64)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, foreach=None,
                 maximize=False, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False):
        # torch.optim.Adam's remaining keyword arguments are accepted at their defaults (a trainer that spells them out
        # keeps working); the variants themselves are not built
        for name, val in (('amsgrad', amsgrad), ('maximize', maximize), ('capturable', capturable),
                          ('differentiable', differentiable), ('decoupled_weight_decay', decoupled_weight_decay)):
            if val:
                raise NotImplementedError('tnp Adam: %s=True is not implemented (use torch.optim.Adam)' % name)
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError('tnp Adam: a tensor lr is not implemented')
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError('invalid Adam hyper-parameters')
        super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad without its per-call bookkeeping (profiler range, foreach grouping; ~50 us of
        host time per optimisation step).  Same effect."""
        if not set_to_none:
            return super(Adam, self).zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None

    # ---- the update -----------------------------------------------------------------------------------------------------
    # A step of the reference's trainer updates ~20 small tensors with ONE launch, so what it costs is the host's walk over
    # them (0.14 ms of a 2 ms batch_size-8 step when every parameter went through validation, a host-tensor `step += 1`, four
    # data_ptr() calls and a new ctypes record).  Parameters that take the update together are kept as a `_Bucket`: the
    # ctypes table with the constant pointers filled in, ONE step tensor shared by their states (same value, so the same
    # state_dict; `state_dict()` hands out per-parameter copies), the list whose version counters move.  Per step and
    # parameter only the gradient is looked at.  Buckets are rebuilt when the set of parameters with a gradient changes, a
    # state is loaded or a group is added.
    class _Bucket(object):
        __slots__ = ('params', 'step', 'table', 'touched', 'device')

    def _build_buckets(self, group, live):
        buckets = {}
        for p in live:
            _lib.require_device(p, 'parameter')
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('tnp Adam: float32 contiguous dense parameters only')
            state = self.state[p]
            if len(state) == 0:
                state['step'] = torch.tensor(0.0, dtype=torch.float32)            # host tensor, like torch's default
                state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            key = (float(state['step']), p.device)
            b = buckets.get(key)
            if b is None:
                b = buckets[key] = Adam._Bucket()
                b.params, b.step, b.device = [], state['step'], p.device
            b.params.append(p)
            state['step'] = b.step                                                # one tensor per bucket
        out = list(buckets.values())
        for b in out:
            b.table = (AdamTensor * len(b.params))()
            b.touched = []
            for i, p in enumerate(b.params):
                st = self.state[p]
                m, v = st['exp_avg'], st['exp_avg_sq']
                if m.device != p.device or v.device != p.device or m.dtype != torch.float32 or v.dtype != torch.float32 \
                        or not m.is_contiguous() or not v.is_contiguous():
                    raise RuntimeError('tnp Adam: optimiser state must be float32, contiguous and on the parameter\'s device')
                b.table[i] = AdamTensor(p.data_ptr(), None, m.data_ptr(), v.data_ptr(), p.numel())
                b.touched += [p, m, v]
        return live, out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        plans = self.__dict__.setdefault('_plans', {})
        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group['params'] if p.grad is not None]
            if not live:
                continue
            plan = plans.get(gi)
            if plan is None or len(plan[0]) != len(live) or any(a is not b for a, b in zip(plan[0], live)):
                plan = plans[gi] = self._build_buckets(group, live)
            lr, (beta1, beta2) = float(group['lr']), group['betas']
            for b in plan[1]:
                table, keep = b.table, None
                for i, p in enumerate(b.params):
                    g = p.grad
                    if g.dtype is not f32 or g.device != b.device or g.is_sparse:
                        # the kernel reads raw float pointers on the parameter's device
                        raise RuntimeError('tnp Adam: gradient is %s%s on %s, parameter is float32 on %s'
                                           % (g.dtype, ' (sparse)' if g.is_sparse else '', g.device, p.device))
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep = (keep or []) + [g]
                    rec = table[i]
                    rec.param, rec.grad = p.data_ptr(), g.data_ptr()
                b.step += 1
                step = int(b.step)
                if torch.cuda.current_device() == b.device.index:
                    rc = L.tnp_adam_step(table, len(b.params), step, lr, float(beta1), float(beta2), float(group['eps']),
                                         float(group['weight_decay']), _lib.stream_ptr())
                else:
                    with torch.cuda.device(b.device):
                        rc = L.tnp_adam_step(table, len(b.params), step, lr, float(beta1), float(beta2), float(group['eps']),
                                             float(group['weight_decay']), _lib.stream_ptr())
                _lib.check(rc, 'tnp_adam_step')
                # the kernel wrote through raw pointers: tell autograd (and everything keyed on Tensor._version, e.g. the
                # re-laid-out copies of the first embedding layer, lstm/lstm.py) that these tensors changed in place
                torch.autograd.graph.increment_version(b.touched)
                del keep
        return loss

    def state_dict(self):
        """torch.optim.Optimizer.state_dict; every parameter gets its OWN copy of the step counter (inside, parameters that are
        updated together share one tensor)."""
        sd = super(Adam, self).state_dict()
        for st in sd['state'].values():
            if isinstance(st.get('step'), torch.Tensor):
                st['step'] = st['step'].clone()
        return sd

    def load_state_dict(self, state_dict):
        super(Adam, self).load_state_dict(state_dict)
        self.__dict__['_plans'] = {}

    def add_param_group(self, param_group):
        super(Adam, self).add_param_group(param_group)
        self.__dict__['_plans'] = {}
