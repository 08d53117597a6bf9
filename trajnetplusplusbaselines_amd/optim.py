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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            by_step = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                _lib.require_device(p, 'parameter')
                if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError('tnp Adam: float32 contiguous dense parameters only')
                if p.grad.dtype != torch.float32 or p.grad.device != p.device:
                    # the kernel reads raw float pointers on the parameter's device
                    raise RuntimeError('tnp Adam: gradient is %s on %s, parameter is float32 on %s'
                                       % (p.grad.dtype, p.grad.device, p.device))
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0, dtype=torch.float32)            # host tensor, like torch's default
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['step'] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault((int(state['step']), p.device), []).append((p, grad, state['exp_avg'], state['exp_avg_sq']))
            for (step, dev), items in by_step.items():
                table = (AdamTensor * len(items))()
                for i, (p, g, m, v) in enumerate(items):
                    table[i] = AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                with torch.cuda.device(dev):
                    _lib.check(L.tnp_adam_step(table, len(items), step, float(group['lr']), float(group['betas'][0]),
                                               float(group['betas'][1]), float(group['eps']), float(group['weight_decay']),
                                               _lib.stream_ptr()), 'tnp_adam_step')
                # the kernel wrote through raw pointers: tell autograd (and everything keyed on Tensor._version, e.g. the
                # re-laid-out copies of the first embedding layer, lstm/lstm.py) that these tensors changed in place
                torch.autograd.graph.increment_version([t for p, _, m, v in items for t in (p, m, v)])
        return loss
