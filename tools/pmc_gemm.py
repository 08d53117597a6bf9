"""Launch a few GEMM-1-shaped linear calls (for rocprofv3 --pmc runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
M, N, K = 2048, 1024, 4096
torch.manual_seed(0)
x = torch.randn(M, K, device='cuda') * (torch.rand(M, K, device='cuda') < 0.12)
w = torch.randn(N, K, device='cuda') / 64
b = torch.zeros(N, device='cuda')
out = torch.empty(M, N, device='cuda')
variants = [int(v) for v in sys.argv[1:]] or [4]
for v in variants:
    for _ in range(3):
        _lib.linear_forward(x, w, b, relu=True, variant=v, out=out)
torch.cuda.synchronize()
