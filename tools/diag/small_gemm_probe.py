"""Tile variants of the step's GEMMs at SMALL M (a single scene / a batch_size-8 training batch): the in-workgroup split-K forms
shorten the per-wave K loop, which is what a handful of workgroups on an empty chip wait for."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from trajnetplusplusbaselines_amd import _lib, synth


def timeit(f, n=300):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, K, what) in ((1024, 4096, 'dense social first layer'), (256, 288, 'config-3 first layer')):
    W = torch.randn(N, K, device='cuda')
    bias = torch.randn(N, device='cuda')
    for M in (2048, 4096, 16384):
        x = torch.randn(M, K, device='cuda')
        out = torch.empty(M, N, device='cuda')
        ref = None
        row = []
        for v in (0, 12, 33):
            try:
                us = timeit(lambda: _lib.linear_forward(x, W, bias, relu=True, variant=v, out=out))
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max())
                row.append('v%d %.1f us (d %.1e)' % (v, us, err))
            except Exception as e:
                row.append('v%d n/a' % v)
        print('%-24s M=%4d N=%4d K=%4d: %s' % (what, M, N, K, '  '.join(row)), flush=True)

