"""Microbenchmarks of host-side torch calls that show up in the small-batch training loop's profile (GPU box: 256 host cores)."""
import time
import numpy as np
import torch


def t(f, n=200):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


print('torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())
sizes = torch.randint(8, 72, (8,))
base = torch.arange(8)
print('repeat_interleave (cpu, 8 -> ~320)     %8.1f us' % t(lambda: torch.repeat_interleave(base, sizes)))
print('np.repeat + from_numpy                 %8.1f us' % t(lambda: torch.from_numpy(np.repeat(base.numpy(), sizes.numpy()))))
torch.set_num_threads(1)
print('repeat_interleave, 1 thread            %8.1f us' % t(lambda: torch.repeat_interleave(base, sizes)))
torch.set_num_threads(16)
print('repeat_interleave, 16 threads          %8.1f us' % t(lambda: torch.repeat_interleave(base, sizes)))
x = torch.zeros(4, device='cuda')
print('empty((), pin_memory=True)             %8.1f us' % t(lambda: torch.empty((), dtype=torch.float32, pin_memory=True)))
buf = torch.empty((), dtype=torch.float32, pin_memory=True)
print('pinned scalar copy_ non_blocking        %8.1f us' % t(lambda: buf.copy_(x[0], non_blocking=True)))
print('Event() + record                        %8.1f us' % t(lambda: torch.cuda.Event().record()))
ps = [torch.nn.Parameter(torch.zeros(1024, 1024, device='cuda')) for _ in range(22)]


def zg():
    for p in ps:
        p.grad = torch.empty_like(p)
    for p in ps:
        p.grad = None


print('22 x (grad = empty_like; grad = None)   %8.1f us' % t(zg))
cpu = torch.arange(300, dtype=torch.int32)
print('small H2D pageable .to(cuda)            %8.1f us' % t(lambda: cpu.to('cuda')))
pin = cpu.pin_memory()
print('pin_memory() of 300 ints                %8.1f us' % t(lambda: cpu.pin_memory()))
print('small H2D pinned non_blocking           %8.1f us' % t(lambda: pin.to('cuda', non_blocking=True)))
print('torch.tensor(list of 9) int64           %8.1f us' % t(lambda: torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8], dtype=torch.int64)))
torch.cuda.synchronize()
