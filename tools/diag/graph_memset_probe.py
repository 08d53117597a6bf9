"""Does a hipMemsetAsync captured into a hipGraph clear the whole range on every replay?  (ROCm 7.2 / torch 2.10)"""
import ctypes
import torch

hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (64, 1000, 4096, 16964, 16968, 65536, 182852, 1 << 20, (1 << 20) + 68):
    buf = torch.full((nbytes + 256,), 7, dtype=torch.uint8, device='cuda')
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        rc = hip.hipMemsetAsync(buf.data_ptr() + 64, 0, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    bad = []
    for rep in range(4):
        buf.fill_(9)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            g.replay()
        torch.cuda.synchronize()
        h = buf.cpu()
        inner, outer = h[64:64 + nbytes], torch.cat([h[:64], h[64 + nbytes:]])
        if int((inner != 0).sum()) or int((outer != 9).sum()):
            bad.append((rep, int((inner != 0).sum()), int((outer != 9).sum())))
    print(nbytes, 'ok' if not bad else bad)
