import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
from tools.gpu_check import time_fn
M, N, K = 2048, 1024, 4096
x = torch.randn(M, K, device='cuda') * (torch.rand(M, K, device='cuda') < 0.12)
w = torch.randn(N, K, device='cuda') / 64
b = torch.zeros(N, device='cuda'); out = torch.empty(M, N, device='cuda')
for rep in range(2):
    for v in (20, 30, 21, 31, 23, 33, 12, 22):
        us = time_fn(lambda: _lib.linear_forward(x, w, b, relu=True, variant=v, out=out), iters=30)
        print('variant', v, '%.1f us  %.1f TF' % (us, 2.0 * M * N * K / us / 1e6))
