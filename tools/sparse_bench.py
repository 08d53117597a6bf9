import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
from tools.gpu_check import time_fn
dev = torch.device('cuda')
B, N, n, C, N1 = 64, 32, 16, 16, 1024
M, ncell = B * N, n * n
g = torch.Generator().manual_seed(0)
SPREAD = float(os.environ.get('SPREAD', '8'))
obs2 = (torch.rand(M, 2, generator=g) * SPREAD - SPREAD / 2).to(dev); obs1 = obs2 - 0.1
enc = torch.randn(M, C, generator=g).to(dev)
st = torch.arange(0, M + 1, N, dtype=torch.int32, device=dev)
W = (torch.randn(N1, C * ncell, generator=g) / 20).to(dev); b = torch.randn(N1, generator=g).to(dev)
Wcm = W.view(N1, C, ncell).permute(2, 1, 0).contiguous()
L = _lib.lib()
winners = torch.empty(M, ncell, dtype=torch.int16, device=dev)
grid = torch.empty(M, C * ncell, device=dev)
_lib.check(L.tnp_pool_grid_forward(2, _lib.ptr(obs1), _lib.ptr(obs2), _lib.ptr(enc), C, _lib.ptr(st), B, N, None, n, C, float(np.float32(0.6)), n / 2, n / 2, 0.0, _lib.ptr(grid), C * ncell, _lib.ptr(winners), _lib.stream_ptr()), 'grid')
print('hits per ego', float((winners >= 0).sum()) / M)
row_base = torch.empty(M, dtype=torch.int32, device=dev)
L.tnp_row_base(_lib.ptr(st), B, _lib.ptr(row_base), _lib.stream_ptr())
need = L.tnp_pool_embed_sparse_workspace_bytes(M, N1, ncell)
ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
out = torch.empty(M, N1, device=dev)
f = lambda: _lib.check(L.tnp_pool_embed_sparse_forward(_lib.ptr(winners), _lib.ptr(enc), C, _lib.ptr(row_base), _lib.ptr(Wcm), _lib.ptr(b), M, ncell, C, N1, 1, _lib.ptr(out), N1, _lib.ptr(ws), need, _lib.stream_ptr()), 'sp')
us = time_fn(f, iters=20)
ref = _lib.linear_forward(grid, W, b, relu=True)
print('TNP_SPARSE_VARIANT', os.environ.get('TNP_SPARSE_VARIANT', '0'), 'sparse %.1f us' % us, 'max err vs dense %.2e' % (out - ref).abs().max().item(),
      'dense %.1f us' % time_fn(lambda: _lib.linear_forward(grid, W, b, relu=True, out=ref), iters=20),
      'grid(winners only) %.1f us' % time_fn(lambda: L.tnp_pool_grid_forward(2, _lib.ptr(obs1), _lib.ptr(obs2), _lib.ptr(enc), C, _lib.ptr(st), B, N, None, n, C, float(np.float32(0.6)), n / 2, n / 2, 0.0, None, 0, _lib.ptr(winners), _lib.stream_ptr()), iters=20))
