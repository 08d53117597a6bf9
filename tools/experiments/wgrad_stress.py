"""Repeat stress of the weight-gradient GEMM's hand-waited operand pipeline (gemm_wgrad.hip): N launches of the same
contraction must be bit-identical (a missing wait would read registers before their loads land -- intermittently)."""
import sys
import torch
sys.path.insert(0, '.')
from trajnetplusplusbaselines_amd import _lib

L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (K, Mo, No) in [(38912, 256, 1024), (16384, 512, 320), (22528, 512, 128), (4096, 128, 128), (48, 64, 64)]:
    g = torch.Generator().manual_seed(K)
    dy, x = torch.randn(K, Mo, generator=g).cuda(), torch.randn(K, No, generator=g).cuda()
    nb = L.tnp_wgrad_workspace_bytes(Mo, No, K)
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    first = None
    diff = 0
    for r in range(reps):
        dw = torch.empty(Mo, No, device='cuda'); db = torch.empty(Mo, device='cuda')
        # other work in flight so that wave timing varies between repeats
        junk = torch.randn(1 << (10 + r % 8), device='cuda').sum()
        _lib.check(L.tnp_wgrad(_lib.ptr(dy), Mo, _lib.ptr(x), No, K, Mo, No, _lib.ptr(dw), No, _lib.ptr(db), _lib.ptr(ws), nb,
                               _lib.stream_ptr()), 'tnp_wgrad')
        if first is None:
            first = (dw.clone(), db.clone())
            want = dy.double().t() @ x.double()
            err = float((dw.double() - want).abs().max() / want.abs().max())
        elif not (torch.equal(dw, first[0]) and torch.equal(db, first[1])):
            diff += 1
    print('K %6d Mo %4d No %4d: max rel err vs fp64 %.2e, %d of %d repeats differ from the first' % (K, Mo, No, err, diff, reps - 1))
    bad += diff
sys.exit(1 if bad else 0)
