import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajnetplusplusbaselines_amd import _lib
L = _lib.lib()
L.tnp_mfma_ablate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
src = torch.randn(256 * 8192 + 16384, device='cuda')
scratch = torch.zeros(4, device='cuda')
iters, blocks = 2000, 256
for mode, sign in [(m, sg) for m in (0, 1, 3, 7) for sg in (1, -1)]:
    for rep in range(2):
        L.tnp_mfma_ablate(mode, 10, blocks, _lib.ptr(src), _lib.ptr(scratch), _lib.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.tnp_mfma_ablate(mode, sign * iters, blocks, _lib.ptr(src), _lib.ptr(scratch), _lib.stream_ptr()), 'ablate')
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = 4096.0 * 16 * iters * 8 * blocks / (ms * 1e-3) / 1e12
    print('random_lds' if sign < 0 else 'pattern_lds', 'mode %d (lds_frag=%d barrier=%d gload+dswrite=%d): %.1f TFLOP/s' % (mode, mode & 1, (mode >> 1) & 1, (mode >> 2) & 1, tf))
