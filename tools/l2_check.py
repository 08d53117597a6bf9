import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
from tools.gpu_check import time_fn
L = _lib.lib()
M, N, K = 2048, 1024, 4096
x = torch.randn(M, K, device='cuda') * (torch.rand(M, K, device='cuda') < 0.12)
w = torch.randn(N, K, device='cuda') / 64
b = torch.zeros(N, device='cuda'); out = torch.empty(M, N, device='cuda')
def run(lda, ldw, v):
    return time_fn(lambda: _lib.check(L.tnp_linear_forward(_lib.ptr(x), lda, _lib.ptr(w), ldw, _lib.ptr(b), _lib.ptr(out), N, M, N, K, 1, v, _lib.stream_ptr()), 'lin'), iters=30)
for v in (20, 22, 12, 4):
    print('variant', v, 'normal %.1f us | A aliased (lda=0) %.1f | W aliased (ldw=0) %.1f | both aliased %.1f' % (run(K, K, v), run(0, K, v), run(K, 0, v), run(0, 0, v)))
