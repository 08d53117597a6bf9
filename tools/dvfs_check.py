import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
from tools.gpu_check import time_fn
M, N, K = 2048, 1024, 4096
out = torch.empty(M, N, device='cuda')
b = torch.zeros(N, device='cuda')
for name, x, w in (('randn', torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') / 64),
                   ('zeros', torch.zeros(M, K, device='cuda'), torch.zeros(N, K, device='cuda')),
                   ('ones', torch.ones(M, K, device='cuda'), torch.ones(N, K, device='cuda')),
                   ('sparse_grid_like', torch.randn(M, K, device='cuda') * (torch.rand(M, K, device='cuda') < 0.12), torch.randn(N, K, device='cuda') / 64)):
    for v in (4, 12, 2):
        us = time_fn(lambda: _lib.linear_forward(x, w, b, relu=True, variant=v, out=out), iters=30)
        print(name, 'variant', v, 'us %.1f' % us, 'TF %.1f' % (2.0 * M * N * K / us / 1e6))
print('probe', _lib.mfma_probe_tflops(8, 1, 4000))
