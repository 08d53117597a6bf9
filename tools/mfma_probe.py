import sys; sys.path.insert(0,'.')
from trajnetplusplusbaselines_amd import _lib
for w in (1,2,4,8):
    for a in (1,2,4):
        print('waves/WG', w, 'acc', a, 'TFLOP/s %.1f' % _lib.mfma_probe_tflops(w, a, 4000))
for it in (200, 1000, 20000):
    print('iters', it, '%.1f' % _lib.mfma_probe_tflops(4, 2, it))
