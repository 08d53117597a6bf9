"""One-shot GPU diagnostics: smoke, per-kernel variant sweep (HIP-event timed) and whole-forward timing.
Prints one JSON line per measurement so a single gpurun call yields everything needed to pick defaults."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from trajnetplusplusbaselines_amd import _lib, synth  # noqa: E402
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling  # noqa: E402


def time_fn(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def sweep_linear():
    shapes = {'gemm1_social': (2048, 1024, 4096), 'gemm2_social': (2048, 256, 1024),
              'gemm1_directional': (2048, 256, 288), 'gemm1_directional_cfg3': (16384, 256, 288)}
    for name, (M, N, K) in shapes.items():
        x = torch.randn(M, K, device='cuda')
        w = torch.randn(N, K, device='cuda') / K ** 0.5
        b = torch.randn(N, device='cuda')
        out = torch.empty(M, N, device='cuda')
        for v in [int(x) for x in os.environ.get('LINEAR_VARIANTS', '0 4 5 12 20 21 22 23 24 28').split()]:
            try:
                us = time_fn(lambda: _lib.linear_forward(x, w, b, relu=True, variant=v, out=out))
            except RuntimeError as e:
                print(json.dumps(dict(kind='linear', shape=name, variant=v, error=str(e))))
                continue
            tf = 2.0 * M * N * K / us / 1e6
            print(json.dumps(dict(kind='linear', shape=name, variant=v, us=round(us, 2), tflops=round(tf, 2),
                                  frac_mfma_peak=round(tf / 157.3, 4))))


def forward_timing():
    torch.manual_seed(0)
    for cfgname, kw, scenes, agents in (
            ('social64x32', dict(type_='social', n=16, out_dim=256, embedding_arch='two_layer', layer_dims=[1024]), 64, 32),
            ('directional32x64', dict(type_='directional', n=12, out_dim=256), 32, 64),
            ('directional256x64', dict(type_='directional', n=12, out_dim=256), 256, 64),
            ('vanilla1x4', None, 1, 4)):
        pool = GridBasedPooling(hidden_dim=128, cell_side=0.6, latent_dim=16, **kw) if kw else None
        model = LSTM(pool=pool).eval().cuda()
        xy, split = synth.linear_crowd(scenes, agents, seed=1)
        obs = xy[:9].cuda()
        goals = torch.zeros(xy.shape[1], 2, device='cuda')
        gate_variants = (0, 5, 20)
        gemm_variants = (0, 1 << 16) if cfgname.startswith('social') else (0,)
        for gv in gate_variants:
            for lv in gemm_variants:
                model.kernel_variant = (lv & 0xff) | (gv << 8) | (lv & (1 << 16))
                try:
                    with torch.no_grad():
                        us = time_fn(lambda: model(obs, goals, split, n_predict=12), iters=10, warmup=2)
                except RuntimeError as e:
                    print(json.dumps(dict(kind='forward', cfg=cfgname, gemm1_variant=lv, gates_variant=gv, error=str(e))))
                    continue
                print(json.dumps(dict(kind='forward', cfg=cfgname, gemm1_variant=lv, gates_variant=gv,
                                      ms=round(us / 1e3, 3), scene_steps_per_s=round(scenes * 21 / (us * 1e-6)))))
        # kernel-class breakdown with the library's event hook (all GEMM launches)
        model.kernel_variant = 0
        L = _lib.lib()
        for which, label in ((0, 'gemm1'), (1, 'all_gemm')):
            L.tnp_profile_begin(which)
            with torch.no_grad():
                model(obs, goals, split, n_predict=12)
            torch.cuda.synchronize()
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            L.tnp_profile_read(ctypes.byref(ms), ctypes.byref(n))
            L.tnp_profile_end()
            print(json.dumps(dict(kind='breakdown', cfg=cfgname, cls=label, total_ms=round(ms.value, 3), launches=n.value)))


if __name__ == '__main__':
    print(json.dumps(dict(kind='device', name=torch.cuda.get_device_name(0),
                          cus=torch.cuda.get_device_properties(0).multi_processor_count,
                          host_cores=os.cpu_count())))
    what = sys.argv[1:] or ['linear', 'forward']
    if 'linear' in what:
        sweep_linear()
    if 'forward' in what:
        forward_timing()
