"""Build libtrajnet_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain
C-ABI shared object (include/trajnet_hip.h); hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT_DIR = os.path.join(PKG, 'lib')
OUT = os.path.join(OUT_DIR, 'libtrajnet_hip.so')
SOURCES = ['gemm_f32_mfma.hip', 'gemm_skinny.hip', 'gemm_wgrad.hip', 'pool_grid.hip', 'pool_embed_sparse.hip', 'pool_nongrid.hip', 'lstm_seq.hip', 'lstm_bwd.hip', 'loss.hip',
           'optim.hip', 'classical.hip', 'ndjson_io.cpp']
EXACT = ('classical.hip', 'pool_grid.hip', 'optim.hip')
HEADERS = ['tnp_internal.h', 'lstm_cell.h', 'classical_core.h', 'grid_build_body.h', os.path.join('..', '..', 'include', 'trajnet_hip.h'),
           os.path.join('..', '..', 'include', 'trajnet_hip_profile.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not (force or needs_build()):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    objs, cmds = [], []
    for src in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(OUT_DIR, os.path.splitext(src)[0] + '.o')
        # exact-arithmetic units (cell indexing, float64 / float32 simulators): no fma contraction
        extra = ['-ffp-contract=off'] if src in EXACT else []
        cmd = [hipcc()] + FLAGS + extra + os.environ.get('TNP_HIPCC_EXTRA', '').split() + ['-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        cmds.append(cmd)
        objs.append(obj)
    # translation units are independent: compile them side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as pool:
        list(pool.map(subprocess.check_call, cmds))
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
