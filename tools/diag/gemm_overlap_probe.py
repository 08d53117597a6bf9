"""Do the two dense GEMMs of the recurrent step (second embedding layer [2048,1024]x[256,1024]^T, gates [2048,448]x[512,448]^T)
run faster when their workgroups share the CUs?  Serial on one stream vs concurrent on two streams (independent operands).
If concurrent ~ max(a, b) rather than a + b, a producer -> consumer fusion at tile granularity (both kernels resident, the
consumer's workgroups waiting on per-tile flags) has room to win; if ~ a + b it has none."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajnetplusplusbaselines_amd import _lib

dev = torch.device('cuda')
M = 2048
x1, w1, b1 = torch.randn(M, 1024, device=dev), torch.randn(256, 1024, device=dev), torch.randn(256, device=dev)
x2, w2, b2 = torch.randn(M, 448, device=dev), torch.randn(512, 448, device=dev), torch.randn(512, device=dev)
y1, y2 = torch.empty(M, 256, device=dev), torch.empty(M, 512, device=dev)
L = _lib.lib()


def launch(x, w, b, y, stream):
    _lib.check(L.tnp_linear_forward(_lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(b), _lib.ptr(y), y.stride(0),
                                    x.shape[0], w.shape[0], x.shape[1], 1, 0, stream), 'linear')


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
p0, p1, p2 = (torch.cuda.current_stream().cuda_stream, s1.cuda_stream, s2.cuda_stream)
import ctypes
c = ctypes.c_void_p
a = timed(lambda: launch(x1, w1, b1, y1, c(p0)))
b = timed(lambda: launch(x2, w2, b2, y2, c(p0)))
ab = timed(lambda: (launch(x1, w1, b1, y1, c(p0)), launch(x2, w2, b2, y2, c(p0))))


def conc():
    launch(x1, w1, b1, y1, c(p1))
    launch(x2, w2, b2, y2, c(p2))


cc = timed(conc)
print('second-layer-shaped GEMM alone      %.2f us' % a)
print('gates-shaped GEMM alone             %.2f us' % b)
print('both, one stream (serial)           %.2f us per pair' % ab)
print('both, two streams (concurrent)      %.2f us per pair' % cc)


def launch_v(x, w, b, y, stream, variant):
    _lib.check(L.tnp_linear_forward(_lib.ptr(x), x.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(b), _lib.ptr(y), y.stride(0),
                                    x.shape[0], w.shape[0], x.shape[1], 1, variant, stream), 'linear')


for v in (24, 27, 25, 26):
    try:
        t = timed(lambda: launch_v(x1, w1, b1, y1, c(p0), v))
        print('second-layer shape, variant %d: %.2f us' % (v, t))
    except Exception as e:
        print('variant', v, 'failed:', str(e)[:100])
