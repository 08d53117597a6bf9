import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
torch.manual_seed(0)
for variant, wk, bk in ((3, 2, 32), (5, 1, 32), (12, 4, 32), (13, 2, 32), (10, 2, 32), (1, 1, 32), (7, 1, 64)):
    for kt in range(1, 8):
        K = wk * bk * kt
        for M, N in ((200, 130), (64, 64)):
            x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda')
            ref = (x.double() @ w.double().t()).float()
            got = _lib.linear_forward(x, w, None, variant=variant)
            err = (got - ref).abs().max().item()
            bad = ((got - ref).abs() > 1e-3)
            msg = ''
            if bad.any():
                rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
                msg = ' BAD rows %s.. cols %s.. n=%d' % (rows[:6], cols[:6], int(bad.sum()))
            print('variant %2d K=%4d KT=%d M=%d N=%d err=%.2e%s' % (variant, K, kt, M, N, err, msg))
