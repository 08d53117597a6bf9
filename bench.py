"""Headline benchmark: scene-steps/s of the Social-LSTM hot path (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one LSTM.forward (8 encoder + 11 decoder recurrent steps = 21 frames) over one batch of synthetic
scenes per GPU (scenes are independent -> pure data-parallel sharding, no collective on the inference data path;
weak scaling).  Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : the dominant kernel (first pooling-embedding layer) timed with HIP events on the launch stream over a
                 repeat of the timed region; `frac` against SURVEY 8(d)'s (A-1)-cell bound and `frac_on_hits` against the
                 occupied cells actually present in the batch (counted on the host from the positions every step ran on),
  training     : the same model's optimisation step (Trainer.train_batch: forward + backward + Adam) on the same shard,
                 timed the same way right after the inference region; at N > 1 it includes the gradient all-reduce over
                 RCCL (parallel.GradBuckets: flat buckets, asynchronous launch) -- the part of the 1 -> 8 GPU curve
                 that can fail to scale (skip with --no-train),
  cpu_baseline : the CPU oracle (a C port of the reference algorithm, OpenMP over tracks) timed on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from trajnetplusplusbaselines_amd import _lib, optim, synth  # noqa: E402
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling  # noqa: E402

CONFIGS = {
    # BASELINE.json configs[1]
    'social': dict(type_='social', n=16, arch='two_layer', layer_dims=[1024], out_dim=256, scenes=64, agents=32,
                   name='Social-LSTM n=16 two_layer 1024'),
    # BASELINE.json configs[2], per-GPU shard of 256 scenes x 64 agents over 8 GPUs
    'directional': dict(type_='directional', n=12, arch='one_layer', layer_dims=None, out_dim=256, scenes=32,
                        agents=64, name='D-LSTM directional n=12 one_layer'),
    # BASELINE.json configs[3], per-GPU shard of 128 scenes x 32 agents over 4 GPUs: SGAN.forward with k = 3 generator
    # samples (noise_dim 16) + real / fake discriminator scores (reference sgan/sgan.py:78-132), inference
    'sgan': dict(type_='directional', n=12, arch='one_layer', layer_dims=None, out_dim=256, scenes=32, agents=32,
                 name='S-GAN directional n=12 k=3 (generator x3 + discriminator x2)', sgan=True),
}
CONFIGS['classical'] = dict(classical=True, scenes=4096, agents=128,
                            name='classical.socialforce + ORCA + Kalman batched rollouts (BASELINE config 5)')
# tests/golden/train_full.npz (oracle/gen_golden_r4.py): the reference's first-step training loss of the headline model
# (default init under seed 123) on synth.linear_crowd(64, 32, seed=100); tests/test_bench_helpers.py checks these against the fixture
TRAIN_PIN_SEED, TRAIN_PIN_LOSS, TRAIN_PIN_RTOL = 123, 149.639572, 2e-5
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
FP64_VALU_PEAK_TFLOPS = 157.3 / 2   # fp64 vector FMA issues at half the fp32 vector rate (= the 78.6 TFLOP/s of AMD's datasheet)


def make_adam(params):
    """Adam(lr 1e-3, weight_decay 1e-4) as the reference's trainers construct it (lstm/trainer.py:497): the native
    one-launch update (optim.Adam = tnp_adam_step) unless TNP_BENCH_TORCH_ADAM=1 asks for torch's foreach kernels."""
    cls = torch.optim.Adam if os.environ.get('TNP_BENCH_TORCH_ADAM') == '1' else optim.Adam
    return cls(params, lr=1e-3, weight_decay=1e-4)


def build_model(cfg, device, seed=0):
    torch.manual_seed(seed)
    if cfg.get('sgan'):
        from trajnetplusplusbaselines_amd.sgan import SGAN, LSTMGenerator, LSTMDiscriminator
        mk = lambda: GridBasedPooling(type_=cfg['type_'], hidden_dim=128, cell_side=0.6, n=cfg['n'], out_dim=cfg['out_dim'],
                                      embedding_arch=cfg['arch'], layer_dims=cfg['layer_dims'], latent_dim=16)
        return SGAN(generator=LSTMGenerator(pool=mk(), noise_dim=16), discriminator=LSTMDiscriminator(pool=mk()),
                    k=3, d_steps=1, g_steps=1).eval().to(device)
    pool = GridBasedPooling(type_=cfg['type_'], hidden_dim=128, cell_side=0.6, n=cfg['n'], out_dim=cfg['out_dim'],
                            embedding_arch=cfg['arch'], layer_dims=cfg['layer_dims'], latent_dim=16)
    return LSTM(pool=pool).eval().to(device)


def step_flops(model, M, agents, sparse_gather=True):
    """Algorithmic FLOPs of ONE recurrent step of `model` over M tracks: first embedding layer (the sparse gather
    bound 2 M (A - 1) C N1 for social grids, the dense GEMM otherwise), the other embedding layers, the LSTM gates, Hidden2Normal /
    social encoding / input embedding.  What `roofline.frac` of every timed leg prices against the fp32 peak."""
    H = model.hidden_dim
    pool = getattr(model, 'pool', None)
    fl = 2.0 * M * (model.encoder.weight_ih.shape[1] + H) * 4 * H + 2.0 * M * H * 5 + 2.0 * M * 2 * 62
    if pool is not None and hasattr(pool, 'embedding_layers'):
        layers = pool.embedding_layers()
        dims = [layers[0].weight.shape[1]] + [l.weight.shape[0] for l in layers]
        social = pool.type_ == 'social'
        C = pool.pooling_dim
        fl += (2.0 * M * (agents - 1) * C * dims[1]) if (social and sparse_gather) else 2.0 * M * dims[0] * dims[1]
        fl += sum(2.0 * M * dims[i] * dims[i + 1] for i in range(1, len(dims) - 1))
        if social:
            fl += 2.0 * M * H * C
    return fl


def leg_roofline(flops, seconds, what):
    return dict(bound='mfma-f32', achieved=flops / seconds / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                frac=flops / seconds / 1e12 / FP32_MFMA_PEAK_TFLOPS, flops=flops, basis=what)


def cpu_baseline(cfg, xy, split, budget_s=20.0):
    """Oracle (kind 'port') on the host cores, same workload, bounded to ~budget_s seconds."""
    import numpy as np
    from oracle import oracle
    torch.manual_seed(0)
    pool = GridBasedPooling(type_=cfg['type_'], hidden_dim=128, cell_side=0.6, n=cfg['n'], out_dim=cfg['out_dim'],
                            embedding_arch=cfg['arch'], layer_dims=cfg['layer_dims'], latent_dim=16)
    sd = {k: v.numpy() for k, v in LSTM(pool=pool).state_dict().items()}
    om = oracle.OracleModel(sd, pool_type=cfg['type_'], n=cfg['n'], cell_side=0.6)
    obs = xy[:9].cpu().numpy()
    sp = split.cpu().numpy()
    scenes = len(sp) - 1
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    max_threads = int(os.environ.get('OMP_NUM_THREADS', cores))
    # The port's OpenMP regions are the per-track Linear layers only (the grid build and the bookkeeping are serial), so it
    # does not scale to a 256-core host: time it at a few thread counts inside the budget and report the BEST one with the
    # thread count it was measured at (`cores`), every (threads, seconds) pair in `thread_sweep`.
    set_threads = getattr(oracle.lib(), 'orc_set_threads', None)
    counts = sorted({max_threads, min(max_threads, 64), min(max_threads, 16), min(max_threads, 8)}, reverse=True)
    if set_threads is None:
        counts = [max_threads]
    sweep, best, best_threads, spent = [], None, max_threads, 0.0
    for nthr in counts:
        if set_threads is not None:
            set_threads(int(nthr))
        t_this = None
        for rep in range(2):                                  # first call of a count also warms its thread pool
            if spent > budget_s and t_this is not None:
                break
            t0 = time.perf_counter()
            om.forward(obs, None, sp, n_predict=12)
            dt = time.perf_counter() - t0
            spent += dt
            t_this = dt if t_this is None else min(t_this, dt)
        sweep.append({'threads': int(nthr), 'seconds_per_forward': t_this})
        if best is None or t_this < best:
            best, best_threads = t_this, int(nthr)
        if spent > budget_s:
            break
    if set_threads is not None:
        set_threads(int(max_threads))
    # `cores` = the thread count the port actually scales to: the smallest one within 5 % of the best time (a 256-thread run
    # that is no faster than the 16-thread run used 16 cores' worth of the host)
    for e in sorted(sweep, key=lambda e: e['threads']):
        if e['seconds_per_forward'] <= 1.05 * best:
            best, best_threads = e['seconds_per_forward'], e['threads']
            break
    return dict(value=scenes * 21 / best, unit='scene-steps/s', cores=best_threads, kind='port',
                sample='full forwards of the same %d-scene batch at %s OpenMP threads (reported: %d threads = the smallest count within 5 %% of the best time), '
                       'oracle/trajnet_oracle.c with OpenMP over tracks in the Linear layers' % (
                           scenes, '/'.join(str(e['threads']) for e in sweep), best_threads),
                seconds_per_forward=best, host_cores=cores, thread_sweep=sweep,
                note='kind "port": the Python reference cannot travel to the GPU box (/root/reference does not exist '
                     'there), so this is the oracle\'s C restatement.  It is SLOWER than the reference itself: the '
                     'reference (PyTorch CPU, 8 vCPUs of the build container) measured 605 scene-steps/s inference and 152 '
                     'per optimisation step while surveying (BASELINE.md section 2) -- quote GPU / CPU ratios against THAT '
                     'figure (reference_python_scene_steps_per_s), not against this port.',
                reference_python_scene_steps_per_s=605.0, reference_python_cores=8)


def occupied_cells_per_step(observed, pred, split, n, cell_side):
    """Occupied grid cells (= hits of the sparse first layer) of every recurrent step, counted on the host from the
    positions the step ran on: encoder step s pools observed[s + 1], decoder step k pools the model's own prediction of
    that frame (n_predict mode, lstm/lstm.py:240-250).  Same fp32 cell arithmetic and last-writer / cell-0 clobber rules
    as the kernels (gridbased_pooling.py:276-293); returns one count per step, summed over all egos."""
    obs, prd, sp = observed.cpu().numpy(), pred.cpu().numpy(), split.cpu().numpy()
    T_obs = obs.shape[0]
    frames = [obs[s + 1] for s in range(T_obs - 1)] + [prd[T_obs - 2 + k] for k in range(prd.shape[0] - (T_obs - 1))]
    cs, half = np.float32(cell_side), np.float32(n / 2)
    counts = []
    for pos in frames:
        pos = np.where(np.isnan(pos).any(axis=1, keepdims=True), np.float32(-500.0), pos).astype(np.float32)
        total = 0
        for lo, hi in zip(sp[:-1], sp[1:]):
            p = pos[lo:hi]
            o = (p[None, :, :] - p[:, None, :]) / cs + half                     # [ego, neighbour, 2]
            inr = ((o >= 0) & (o < n)).all(axis=2)
            np.fill_diagonal(inr, False)
            cell = np.where(inr, o[..., 0].astype(np.int64) * n + o[..., 1].astype(np.int64), -1)
            for e in range(hi - lo):
                js = np.nonzero(inr[e])[0]
                occupied = set(cell[e, js].tolist())
                oor = np.nonzero(~inr[e])[0]
                oor = oor[oor != e]
                if 0 in occupied and oor.size and oor.max() > js[cell[e, js] == 0].max():
                    occupied.discard(0)                                          # clobbered by a later out-of-range neighbour
                total += len(occupied)
        counts.append(total)
    return counts


def cpu_baseline_classical(st, pos, vel, goals, speed, obs, z, starts, agents, scenes_sample, scenes_total):
    """cpu_baseline leg of the classical rollouts (tools/bench_classical.py, BASELINE config 5): the oracle's host
    execution of the same arithmetic (OpenMP over scenes) on a bounded sample of scenes, extrapolated to the full batch."""
    from oracle import oracle
    n, A, scale = scenes_sample, agents, scenes_total / float(scenes_sample)
    cpu = {}
    t0 = time.perf_counter(); oracle.sf_rollout(st[:n * A], starts[:n + 1]); cpu['socialforce'] = (time.perf_counter() - t0) * scale
    t0 = time.perf_counter()
    oracle.orca_rollout(pos[:n * A], vel[:n * A], goals[:n * A], speed[:n * A], 1.3 * speed[:n * A], starts[:n + 1])
    cpu['orca'] = (time.perf_counter() - t0) * scale
    t0 = time.perf_counter(); oracle.kalman_predict(obs[:n * A], z[:n * A]); cpu['kalman'] = (time.perf_counter() - t0) * scale
    return cpu


def bench_classical(args, cfg, device):
    """BASELINE config 5: social force, ORCA and Kalman rollouts of 4096 scenes x 128 agents (9 obs + 12 pred), inputs
    resident in HBM, one "step" = all three predictors over the whole batch (one workgroup per scene, state in LDS).
    `roofline` prices the dominant predictor (social force, float64) against the fp64 vector peak: SURVEY 8(d) counts ~107
    floating-point operations per ordered pair per simulation step (three evaluations of the elliptical potential with
    2 sqrt + 1 exp each, the finite-difference gradient, the field-of-view test) -- counting sqrt / exp / divide as one
    operation each, which flatters nothing: they cost several issue slots on the hardware."""
    S, A = cfg['scenes'], cfg['agents']
    rng = np.random.RandomState(11)
    M = S * A
    pos = rng.rand(M, 2) * 8 - 4
    vel = rng.randn(M, 2) * 0.6
    goals = pos + vel * 4.8 + rng.randn(M, 2) * 0.2
    speed = np.linalg.norm(vel, axis=1)
    st = np.concatenate([pos, vel, goals], axis=1)
    starts = np.arange(S + 1, dtype=np.int64) * A
    t = np.arange(9)[None, :, None]
    obs = pos[:, None, :] + vel[:, None, :] * 0.4 * (t - 8) + rng.randn(M, 9, 2) * 0.03
    z = rng.standard_normal((M, 5, 13, 6))
    dv = lambda a, dt: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=device)
    d = dict(st=dv(st, np.float64), starts=dv(starts, np.int32), pos=dv(pos, np.float32), vel=dv(vel, np.float32),
             goals=dv(goals, np.float64), speed=dv(speed, np.float64), vmax=dv(1.3 * speed, np.float32), obs=dv(obs, np.float64),
             z=dv(z, np.float64))
    out_sf = torch.empty(12, M, 2, dtype=torch.float64, device=device)
    out_orca = torch.empty(12, M, 2, dtype=torch.float32, device=device)
    out_k = torch.empty(M, 13, 2, dtype=torch.float64, device=device)
    L, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def step(timed=False):
        if timed:
            ev[0].record()
        _lib.check(L.tnp_sf_rollout(P(d['st']), P(d['starts']), S, M, A, 12 * 8, 8, 0.5, 2.1, 0.3, 1.0 / 20, P(out_sf), sp()), 'sf')
        if timed:
            ev[1].record()
        _lib.check(L.tnp_orca_rollout(P(d['pos']), P(d['vel']), P(d['goals']), P(d['speed']), P(d['vmax']), P(d['starts']), S, M, A,
                                      8 * 12 + 1, 8, 1.0 / 20, 1.5, 10, 1.5, 0.4, P(out_orca), None, sp()), 'orca')
        if timed:
            ev[2].record()
        _lib.check(L.tnp_kalman_predict(P(d['obs']), M, 9, 10, 13, 5, P(d['z']), 1e-5, 0.05 ** 2, P(out_k), sp()), 'kalman')
        if timed:
            ev[3].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step(timed=True)
    torch.cuda.synchronize()
    per = {k: ev[i].elapsed_time(ev[i + 1]) for i, k in enumerate(('socialforce', 'orca', 'kalman'))}
    out = {'metric': 'scene-steps/sec (9 obs + 12 pred)', 'value': S * 21 * args.steps / elapsed, 'unit': 'scene-steps/s',
           'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64 (social force, Kalman) / f32 (ORCA)',
           'data': 'synthetic',
           'config': {'workload': '%s, %d scenes x %d agents; one step = all three predictors over the batch' % (cfg['name'], S, A),
                      'ms_per_predictor': per,
                      'scene_steps_per_s_per_predictor': {k: S * 21 / (v * 1e-3) for k, v in per.items()}},
           'roofline': None, 'cpu_baseline': None}
    sf_ops = float(S) * A * (A - 1) * (12 * 8) * 107.0
    sf_s = per['socialforce'] * 1e-3
    out['roofline'] = {'bound': 'valu-f64', 'achieved': sf_ops / sf_s / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                       'frac': sf_ops / sf_s / 1e12 / FP64_VALU_PEAK_TFLOPS, 'traffic': None,
                       'kernel': 'sf_rollout_kernel (one workgroup per scene, state in LDS, 96 Euler steps fused)',
                       'ops_per_launch': sf_ops, 'launch_ms': per['socialforce'],
                       'note': 'operations = scenes x A(A-1) ordered pairs x 96 steps x ~107 (3 potential evaluations with 2 sqrt + 1 exp '
                               'each, gradient, field-of-view test; sqrt / exp / divide counted as ONE operation); HBM traffic is '
                               'the 6 input and 24 output doubles per agent (63 MB per batch) -- irrelevant next to the arithmetic'}
    if not args.no_cpu_baseline:
        sub_n = 64
        cpu = cpu_baseline_classical(st, pos, vel, goals, speed, obs, z, starts, A, sub_n, S)
        tot = sum(cpu.values())
        out['cpu_baseline'] = {'value': S * 21 / tot, 'unit': 'scene-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
                               'sample': '%d of %d scenes per predictor, extrapolated (oracle/classical_oracle.c, OpenMP over scenes)' % (sub_n, S),
                               'seconds_per_predictor_extrapolated': cpu}
    print(json.dumps(out))


def under_profiler():
    """True when this process already runs under rocprofv3 / a rocprofiler tool (no nested PMC child runs then)."""
    if any(k.startswith(('ROCPROF', 'ROCPROFILER', 'ROCP_')) for k in os.environ):
        return True
    return 'rocprof' in os.environ.get('LD_PRELOAD', '') or 'rocprof' in os.environ.get('HSA_TOOLS_LIB', '')


def pmc_traffic(kernel_regex, extra_args):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected as the microarchitecture guide
    prescribes: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), counters + kernel trace only, a short
    child run of this script; FETCH_SIZE is reported in KiB and under-reports wide coalesced reads by 2x on gfx950
    (guide, HBM section) -- both corrections are applied here and the raw values are returned alongside."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    per_pass_timeout = 150
    here = os.path.abspath(__file__)
    raw = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix='tnp_pmc_', dir='/tmp')
        try:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', tmp, '-o', 'p', '--',
                   sys.executable, here, '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-roofline', '--no-traffic'] + extra_args
            subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp', TNP_BENCH_PRIME_S='0'), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=per_pass_timeout, check=False)
            vals = []
            for f in glob.glob(tmp + '/**/*counter_collection.csv', recursive=True):
                with open(f, newline='') as fh:
                    for row in csv.DictReader(fh):
                        if row.get('Counter_Name') == counter and re.search(kernel_regex, row.get('Kernel_Name', '')):
                            vals.append(float(row['Counter_Value']))
            if not vals:
                return None
            raw[counter] = sum(vals) / len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch = raw['FETCH_SIZE'] * 1024.0 * 2.0
    write = raw['WRITE_SIZE'] * 1024.0
    return dict(bytes=fetch + write, fetch_bytes=fetch, write_bytes=write, raw_fetch_kib=raw['FETCH_SIZE'],
                raw_write_kib=raw['WRITE_SIZE'],
                note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE x2 (gfx950 '
                     'wide-read correction of the microarchitecture guide); mean per launch of the dominant kernel')


def rocprof_kernel_us(kernel_regex, extra_args, steps=20):
    """Mean duration (us) of the dominant kernel by rocprofv3's own kernel trace (no counters): a short child run of this
    script under `rocprofv3 --kernel-trace`, so that the line carries the profiler's number beside the HIP-event one
    (the two were 9 % apart on the driver's box).  None when rocprofv3 is missing / already profiling."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    tmp = tempfile.mkdtemp(prefix='tnp_kt_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', tmp, '-o', 'k', '--', sys.executable,
               os.path.abspath(__file__), '--steps', str(steps), '--warmup', '3', '--no-cpu-baseline', '--no-roofline', '--no-traffic'] + extra_args
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp', TNP_BENCH_PRIME_S='0.3'), stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=200, check=False)
        durs = []
        for f in glob.glob(tmp + '/**/*kernel_trace.csv', recursive=True):
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    if re.search(kernel_regex, row.get('Kernel_Name', '')):
                        durs.append((int(row['Start_Timestamp']), int(row['End_Timestamp']) - int(row['Start_Timestamp'])))
        if len(durs) < 40:
            return None
        durs.sort()
        tail = [d for _, d in durs[len(durs) // 2:]]          # the second half of the run: primed, clocks up
        return dict(avg_launch_us=sum(tail) / len(tail) / 1e3, launches=len(tail))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def strong_scaling_leg(args, device, rank, world, barrier, global_scenes=256):
    """BASELINE config 3 as STRONG scaling inside the default run: every rank builds the same batch of 256 scenes x 64
    agents (D-LSTM directional n=12 one_layer), keeps parallel.shard_batch's shard (batch-wide slot and scene counts) and
    times the inference forward and the optimisation step (gradient all-reduce from inside the backward pass at N > 1) with
    the contract's barrier + MAX-over-ranks rule.  value = 256 x 21 x steps / time, whatever N is."""
    import torch.distributed as dist
    from trajnetplusplusbaselines_amd import parallel
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    cfg = CONFIGS['directional']
    model = build_model(cfg, device)
    gxy, gsplit = synth.linear_crowd(global_scenes, cfg['agents'], seed=100)
    shard = parallel.shard_batch(gxy, torch.zeros(gxy.shape[1], 2), gsplit, rank, world)
    lo, hi = shard.track_range
    xy, split = gxy[:, lo:hi].contiguous(), shard.batch_split
    observed = xy[:9].to(device)
    goals = torch.zeros(xy.shape[1], 2, device=device)

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    steps = max(5, min(args.steps, 50))
    with torch.no_grad():
        t_inf = timed(lambda: model(observed, goals, split, n_predict=12, pad_to=shard.pad_to), steps, 3)
    tmodel = build_model(cfg, device)
    optimizer = make_adam(tmodel.parameters())
    criterion = PredictionLoss()
    scene_dev = xy.to(device)
    t_steps = max(5, min(args.steps, 20))
    with torch.enable_grad():
        t_tr = timed(lambda: train_batch(tmodel, optimizer, criterion, scene_dev, goals, split, 9, 12, batch_size=shard.n_scenes_global,
                                         n_global_scenes=shard.n_scenes_global, pad_to=shard.pad_to, overlap=True), t_steps, 3)
    grad_bytes = sum(p.numel() * 4 for p in tmodel.parameters() if p.grad is not None)
    # roofline of both legs: algorithmic FLOPs of the shard this rank ran (19 recurrent steps; training = forward + a backward of
    # twice the forward's multiply-adds) over its time; the ranks run concurrently, so the fraction is per GPU
    fl = step_flops(model, xy.shape[1], cfg['agents']) * 19
    predicted = None
    if world == 1 and rank == 0:
        # What an 8-GPU run of this leg can reach at best: the time of ONE rank's shard (32 of the 256 scenes, padded as the
        # full batch pads) against the time of the whole batch here -- kernels of a small shard do not shrink in proportion
        # (launch floors), which is what caps strong scaling before any communication does.  Training adds the gradient
        # all-reduce, not modelled: an upper bound.
        sh8 = parallel.shard_batch(gxy, torch.zeros(gxy.shape[1], 2), gsplit, 0, 8)
        lo8, hi8 = sh8.track_range
        x8 = gxy[:, lo8:hi8].contiguous().to(device)
        g8 = torch.zeros(x8.shape[1], 2, device=device)
        with torch.no_grad():
            t8_inf = timed(lambda: model(x8[:9], g8, sh8.batch_split, n_predict=12, pad_to=sh8.pad_to), steps, 3)
        with torch.enable_grad():
            t8_tr = timed(lambda: train_batch(tmodel, optimizer, criterion, x8, g8, sh8.batch_split, 9, 12, batch_size=sh8.n_scenes_global,
                                              pad_to=sh8.pad_to), t_steps, 3)
        predicted = dict(shard='%d of %d scenes' % (sh8.n_scenes, global_scenes), inference_ms_shard=t8_inf / steps * 1e3,
                         training_ms_shard=t8_tr / t_steps * 1e3, inference_speedup_upper_bound=t_inf / t8_inf * steps / steps,
                         training_speedup_upper_bound=(t_tr / t_steps) / (t8_tr / t_steps),
                         note='time of the whole batch on this GPU / time of one of eight shards on this GPU; north_star asks for >= 6x at 8 GPUs')
    return dict(workload='%s, ONE batch of %d scenes x %d agents x (9 obs + 12 pred) sharded over %d GPU(s)' % (
                    cfg['name'], global_scenes, cfg['agents'], world),
                scaling='strong', n_gpus=world, global_scenes=global_scenes, scenes_this_rank=shard.n_scenes,
                inference=dict(value=global_scenes * 21 * steps / t_inf, unit='scene-steps/s', steps=steps, ms_per_step=t_inf / steps * 1e3,
                               roofline=leg_roofline(fl, t_inf / steps, 'one forward of this rank\'s shard: 19 recurrent steps, dense first layer')),
                training=dict(value=global_scenes * 21 * t_steps / t_tr, unit='scene-steps/s', steps=t_steps, ms_per_step=t_tr / t_steps * 1e3,
                              allreduce_bytes=grad_bytes if world > 1 else 0,
                              overlap='in-backward (parallel.GradReducer)' if world > 1 else None,
                              roofline=leg_roofline(3 * fl, t_tr / t_steps, 'forward + data and weight gradients = 3 x the forward\'s FLOPs')),
                predicted_8gpu=predicted)


def sgan_strong_leg(args, device, rank, world, barrier, global_scenes=128):
    """BASELINE config 4 as a strong-scaling leg of the default run: S-GAN (directional n=12 generator + discriminator, k = 3,
    noise_dim 16) on ONE batch of 128 scenes x 32 agents sharded over the ranks; SGAN.forward (teacher-forced 'g' mode: 3
    generator samples + real / fake scores) and one discriminator step + one generator step (variety loss + adversarial loss,
    gradients of the updated network SUM-reduced over the ranks, sgan/train_step.py)."""
    import random
    import torch.distributed as dist
    from trajnetplusplusbaselines_amd import parallel
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    from trajnetplusplusbaselines_amd.sgan.train_step import train_batch as sgan_train_batch
    cfg = CONFIGS['sgan']
    model = build_model(cfg, device)
    gxy, gsplit = synth.linear_crowd(global_scenes, cfg['agents'], seed=44)
    shard = parallel.shard_batch(gxy, torch.zeros(gxy.shape[1], 2), gsplit, rank, world)
    lo, hi = shard.track_range
    xy, split = gxy[:, lo:hi].contiguous().to(device), shard.batch_split
    goals = torch.zeros(xy.shape[1], 2, device=device)
    torch.manual_seed(1234)          # the same noise vectors and noisy labels on every rank
    random.seed(1234)

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    steps = max(5, min(args.steps, 30))
    model.eval()
    with torch.no_grad():
        t_inf = timed(lambda: model(xy[:9], goals, split, prediction_truth=xy[9:21], step_type='g', pred_length=12,
                                    pad_to=shard.pad_to), steps, 3)
    model.train()
    g_opt, d_opt = make_adam(model.generator.parameters()), make_adam(model.discriminator.parameters())
    crit = PredictionLoss(keep_batch_dim=True)
    t_steps = max(3, min(args.steps, 10))

    def dg():
        for st in ('d', 'g'):
            sgan_train_batch(model, g_opt, d_opt, crit, xy, goals, split, st, n_global_scenes=shard.n_scenes_global,
                             pad_to=shard.pad_to)
    with torch.enable_grad():
        t_tr = timed(dg, t_steps, 2)
    rec = 3 * 19 + 2 * 20            # recurrent steps of one SGAN.forward: 3 generator samples + 2 discriminator encodings
    Ml = xy.shape[1]
    fl_fwd = step_flops(model.generator, Ml, cfg['agents']) * 3 * 19 + step_flops(model.discriminator, Ml, cfg['agents']) * 2 * 20
    # one d step + one g step: the d step runs the generator without gradients (3 x 19 steps) and trains the discriminator on
    # real + fake (2 x 20 steps, x 3 with its backward); the g step trains the generator (3 x 19 steps x 3) through the
    # discriminator's fake score (20 steps forward + data gradient: x 2)
    fl_dg = (step_flops(model.generator, Ml, cfg['agents']) * 3 * 19 * (1 + 3) +
             step_flops(model.discriminator, Ml, cfg['agents']) * 20 * (2 * 3 + 2))
    return dict(workload='%s, ONE batch of %d scenes x %d agents sharded over %d GPU(s)' % (cfg['name'], global_scenes, cfg['agents'], world),
                scaling='strong', n_gpus=world, global_scenes=global_scenes, scenes_this_rank=shard.n_scenes,
                recurrent_steps_per_forward=rec,
                inference=dict(value=global_scenes * 21 * steps / t_inf, unit='scene-steps/s (one SGAN.forward = 21 frames per scene)',
                               steps=steps, ms_per_step=t_inf / steps * 1e3,
                               roofline=leg_roofline(fl_fwd, t_inf / steps, '3 x 19 generator steps + 2 x 20 discriminator steps of this rank\'s shard')),
                training=dict(value=global_scenes * 21 * t_steps / t_tr, unit='scene-steps/s (one d step + one g step = 21 frames per scene)',
                              steps=t_steps, ms_per_step=t_tr / t_steps * 1e3,
                              allreduce_bytes=(sum(p.numel() * 4 for p in model.parameters()) if world > 1 else 0),
                              roofline=leg_roofline(fl_dg, t_tr / t_steps, 'd step (generator forward, discriminator forward + backward) + g step '
                                                                           '(generator forward + backward through the discriminator\'s score)')))


def classical_leg(device, scenes=4096, agents=128):
    """BASELINE config 5 inside the default run (N = 1): one timed pass of the three classical rollouts over 4096 scenes x
    128 agents, inputs resident in HBM (bench.py --config classical has the full line with roofline and cpu_baseline)."""
    S, A = scenes, agents
    rng = np.random.RandomState(11)
    M = S * A
    pos = rng.rand(M, 2) * 8 - 4
    vel = rng.randn(M, 2) * 0.6
    goals = pos + vel * 4.8 + rng.randn(M, 2) * 0.2
    speed = np.linalg.norm(vel, axis=1)
    t = np.arange(9)[None, :, None]
    obs = pos[:, None, :] + vel[:, None, :] * 0.4 * (t - 8) + rng.randn(M, 9, 2) * 0.03
    z = rng.standard_normal((M, 5, 13, 6))
    dv = lambda a, dt: torch.tensor(np.ascontiguousarray(a, dtype=dt), device=device)
    d = dict(st=dv(np.concatenate([pos, vel, goals], axis=1), np.float64), starts=dv(np.arange(S + 1) * A, np.int32),
             pos=dv(pos, np.float32), vel=dv(vel, np.float32), goals=dv(goals, np.float64), speed=dv(speed, np.float64),
             vmax=dv(1.3 * speed, np.float32), obs=dv(obs, np.float64), z=dv(z, np.float64))
    out_sf = torch.empty(12, M, 2, dtype=torch.float64, device=device)
    out_orca = torch.empty(12, M, 2, dtype=torch.float32, device=device)
    out_k = torch.empty(M, 13, 2, dtype=torch.float64, device=device)
    L, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for timed in (False, True):
        ev[0].record()
        _lib.check(L.tnp_sf_rollout(P(d['st']), P(d['starts']), S, M, A, 12 * 8, 8, 0.5, 2.1, 0.3, 1.0 / 20, P(out_sf), sp()), 'sf')
        ev[1].record()
        _lib.check(L.tnp_orca_rollout(P(d['pos']), P(d['vel']), P(d['goals']), P(d['speed']), P(d['vmax']), P(d['starts']), S, M, A,
                                      8 * 12 + 1, 8, 1.0 / 20, 1.5, 10, 1.5, 0.4, P(out_orca), None, sp()), 'orca')
        ev[2].record()
        _lib.check(L.tnp_kalman_predict(P(d['obs']), M, 9, 10, 13, 5, P(d['z']), 1e-5, 0.05 ** 2, P(out_k), sp()), 'kalman')
        ev[3].record()
        torch.cuda.synchronize()
    per = {k: ev[i].elapsed_time(ev[i + 1]) for i, k in enumerate(('socialforce', 'orca', 'kalman'))}
    tot = sum(per.values())
    # Roofline of each rollout on an INSTRUCTION-ISSUE basis: these kernels touch 63 MB of HBM per batch and run no matrix
    # instruction; what bounds them is the vector pipe's issue rate -- one wave64 VALU instruction per 4 cycles per SIMD (fp32 and
    # fp64 FMA alike), 1024 SIMDs.  `achieved` = vector instructions per launch (SQ_INSTS_VALU, a property of the workload:
    # profiles/round6_pmc_classical.md, this very batch) x 64 lanes / the launch time measured HERE; `peak` = 1024 SIMDs x 16
    # lanes x the boost clock.  frac = share of the launch during which the vector pipes issue.
    roof = None
    if (S, A) == (4096, 128):
        peak = 1024 * 16 * BOOST_CLOCK_GHZ * 1e9
        roof = {}
        for k, n_instr in CLASSICAL_VALU_INSTS.items():
            ach = n_instr * 64 / (per[k] * 1e-3)
            roof[k] = {'bound': 'valu-issue', 'achieved': ach / 1e12, 'peak': peak / 1e12, 'unit': 'T lane-ops/s', 'frac': ach / peak,
                       'valu_instructions_per_launch': n_instr, 'launch_ms': per[k],
                       'source': 'SQ_INSTS_VALU of profiles/round6_pmc_classical.md (rocprofv3 --pmc over bench.py --config classical)'}
    return dict(workload='classical.socialforce + ORCA + Kalman, %d scenes x %d agents, 9 obs + 12 pred, second pass timed' % (S, A),
                roofline=roof,
                ms_per_predictor=per, scene_steps_per_s_per_predictor={k: S * 21 / (v * 1e-3) for k, v in per.items()},
                value=S * 21 / (tot * 1e-3), unit='scene-steps/s (all three predictors over the batch)',
                finite=bool(torch.isfinite(out_sf).all() and torch.isfinite(out_orca).all() and torch.isfinite(out_k).all()))


# vector instructions per launch at BASELINE config 5 (4096 x 128), profiles/round6_pmc_classical.md; ORCA before round 6's
# register form: 4.628e10 (profiles/archive/round3_d_pmc_classical.md)
CLASSICAL_VALU_INSTS = {'socialforce': 2.297e10, 'orca': 5.416e9, 'kalman': 1.084e9}     # (start of round 6: 3.636e10 / 4.628e10 / 1.492e9)
BOOST_CLOCK_GHZ = 2.4

# the Python reference at its real operating point, measured in the build container (8 vCPUs) by tools/ref_operating_point.py
REF_TRAINER_DEFAULT_MS, REF_PER_SCENE_MS = 1418.4, 81.2      # (a second run read 1249.1 / 86.4)


def operating_point_legs(device, steps=200):
    """The reference's REAL operating point (VERDICT r4 "missing 3"), reported beside the headline -- never as `value`:

    trainer_default   scripts/interaction/social.sh trains the headline model at the trainer's default batch_size = 8
                      (lstm/trainer.py:30,125) on ragged scenes with a NEW batch_split every step.  Here: the headline model, 8
                      scenes per step drawn fresh from a pool of 256 ragged scenes (8..72 agents, ~20 % NaN entries: SURVEY 8d
                      regime ii) through data.SceneBatcher with the rotation augmentation, `steps` optimisation steps, nothing
                      cached across steps that a trainer would not have (scene tables are rebuilt for every new split).
    per_scene_predict LSTMPredictor.__call__ on single scenes (the evaluator's call, lstm/lstm.py:285-313): host paths in,
                      numpy predictions out, one blocking call per scene; and the same scenes through predict_batch, 64 at a
                      time (data.predict_dataset's default)."""
    import random as _random
    from oracle import oracle as _oracle
    from trajnetplusplusbaselines_amd import _lib, data as trajdata
    from trajnetplusplusbaselines_amd.lstm import LSTMPredictor, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    cfg = CONFIGS['social']
    out = {}
    # ---- trainer_default ----
    model = build_model(cfg, device, seed=1)
    optimizer = make_adam(model.parameters())
    criterion = PredictionLoss()
    xy, split = synth.ragged_crowd(256, 8, 72, seed=2024, nan_frac=0.2)
    xy_np, split_np = xy.numpy(), split.numpy()
    scenes = [xy_np[:, split_np[i]:split_np[i + 1]] for i in range(len(split_np) - 1)]
    batcher = trajdata.SceneBatcher(scenes, device=device, drop_distant_r=None)
    rng = _random.Random(7)
    order = [[rng.randrange(len(scenes)) for _ in range(8)] for _ in range(steps + 10)]
    warm = [[rng.randrange(len(scenes)) for _ in range(8)] for _ in range(60)]

    def one(ids):
        bxy, bgoals, bsplit = batcher.batch(ids, augment=True)
        return train_batch(model, optimizer, criterion, bxy, bgoals, bsplit, 9, 12, batch_size=8)

    for ids in order[:10] + warm:          # ragged shapes: the caching allocator has seen the size classes before anything is timed
        one(ids)
    torch.cuda.synchronize()

    def timed_loop():
        _lib.SceneIndex._cache.clear()
        t0 = time.perf_counter()
        n_tracks = 0
        for ids in order[10:]:
            one(ids)
            n_tracks += int(sum(batcher.sizes[ids]))
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return th, time.perf_counter() - t0, n_tracks

    timed_loop()
    t_host, t_all, tracks = timed_loop()
    # the same loop with Python's cyclic collector kept off torch's long-lived objects (gc.freeze(): one line in a trainer):
    # a full collection walks every module object torch created at import, and the step's many small host objects trigger it
    import gc
    gc.collect()
    gc.freeze()
    try:
        t_host_f, t_all_f, _ = timed_loop()
    finally:
        gc.unfreeze()
    # device time of a step in isolation: events around a step with the queue drained in front of it
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for (a, b), ids in zip(ev, order[10:50]):
        torch.cuda.synchronize()
        a.record()
        one(ids)
        b.record()
    torch.cuda.synchronize()
    gpu_ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    out['trainer_default'] = dict(
        workload='Social-LSTM n=16 two_layer 1024 (the headline model), batch_size 8, ragged scenes of 8..72 agents with ~20 %% '
                 'NaN entries, a new batch_split and rotation every step (data.SceneBatcher), %d optimisation steps '
                 '(forward, NLL loss, backward, Adam)' % steps,
        ms_per_step=t_all / steps * 1e3, host_enqueue_ms_per_step=t_host / steps * 1e3, gpu_ms_per_step_isolated=gpu_ms,
        ms_per_step_gc_frozen=t_all_f / steps * 1e3, host_enqueue_ms_per_step_gc_frozen=t_host_f / steps * 1e3,
        tracks_per_step=tracks / steps, scene_steps_per_s=8 * 21 * steps / t_all, steps=steps,
        reference_python_ms_per_step=REF_TRAINER_DEFAULT_MS,
        reference_note='the Python reference on the same crowd and batch size (forward + loss + backward + Adam as Trainer.train_batch), '
                       '8 vCPUs of the build container: tools/ref_operating_point.py -> profiles/round5_reference_operating_point.txt '
                       '(it cannot run on the GPU box)',
        note='ms_per_step: wall clock of the pipelined loop; host_enqueue: until the loop has queued its last step (the host '
             'only blocks on queue back-pressure); gpu isolated: median HIP-event time of one step started on an idle queue '
             '(host enqueue included where the device waits for it); gc_frozen: the same loop after gc.freeze() -- Python\'s '
             'cyclic collector otherwise walks torch\'s import-time objects every few steps (INTEGRATION.md)')
    del model, optimizer
    # ---- per_scene_predict ----
    model = build_model(cfg, device, seed=1).eval()
    predictor = LSTMPredictor(model)
    n_sc = 128
    paths_list = [trajdata.xy_to_paths(sc) for sc in scenes[:n_sc]]
    goals_list = [np.zeros((sc.shape[1], 2)) for sc in scenes[:n_sc]]
    import gc
    gc.collect()                      # the training leg's garbage is not the evaluator's
    t_call = None
    for _ in range(2):                # first pass: the device allocator meets the scenes' sizes; reported: the second
        _lib.SceneIndex._cache.clear()
        t0 = time.perf_counter()
        for paths, g in zip(paths_list, goals_list):
            predictor(paths, g, n_predict=12)
        t_call = (time.perf_counter() - t0) / n_sc
    # device part of one call: the model forward alone on resident tensors, events on an idle queue
    dev_ms = []
    for sc in scenes[:32]:
        obs = torch.tensor(sc[:9], dtype=torch.float32, device=device)
        gl = torch.zeros(sc.shape[1], 2, device=device)
        sp = torch.tensor([0, sc.shape[1]])
        with torch.no_grad():
            model(obs, gl, sp, n_predict=12)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model(obs, gl, sp, n_predict=12)
            b.record()
            torch.cuda.synchronize()
        dev_ms.append(a.elapsed_time(b))
    t0 = time.perf_counter()
    for lo in range(0, n_sc, 64):
        predictor.predict_batch(list(zip(paths_list[lo:lo + 64], goals_list[lo:lo + 64])), n_predict=12)
    t_batch = (time.perf_counter() - t0) / n_sc
    # CPU beside it: the oracle's C port of the same forward on the same single scenes (bounded sample)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    om = _oracle.OracleModel(sd, pool_type='social', n=cfg['n'], cell_side=0.6)
    t0 = time.perf_counter()
    n_cpu = 0
    for sc in scenes[:n_sc]:
        om.forward(sc[:9].astype(np.float32), None, np.array([0, sc.shape[1]]), n_predict=12)
        n_cpu += 1
        if time.perf_counter() - t0 > 8.0:
            break
    t_cpu = (time.perf_counter() - t0) / n_cpu
    out['per_scene_predict'] = dict(
        workload='LSTMPredictor.__call__ (the evaluator\'s per-scene call) on %d single scenes of 8..72 agents: host paths in, '
                 'numpy predictions out, one blocking call per scene' % n_sc,
        ms_per_call=t_call * 1e3, device_ms_per_forward=sorted(dev_ms)[len(dev_ms) // 2], scenes_per_s=1.0 / t_call,
        predict_batch_64_ms_per_scene=t_batch * 1e3, predict_batch_scenes_per_s=1.0 / t_batch,
        cpu_port_ms_per_scene=t_cpu * 1e3, reference_python_ms_per_scene=REF_PER_SCENE_MS, cpu_port_note='oracle/trajnet_oracle.c forward of the same scenes on the host (kind "port"); the '
                                                         'Python reference needs 6.4 ms for a 1 x 4 VANILLA forward (BASELINE.md section 2) '
                                                         'and ~2 s for the 64 x 32 social batch',
        mean_agents=float(np.mean([sc.shape[1] for sc in scenes[:n_sc]])))
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run this script under `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` (what the contract's command line does) and
    exit with the launcher's status; rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='social', choices=sorted(CONFIGS))
    ap.add_argument('--variant', type=int, default=0, help='kernel variant selector (DESIGN.md)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--train', action='store_true', help='time one optimisation step (forward + backward + Adam + gradient all-reduce) instead of the inference forward')
    ap.add_argument('--dense', action='store_true', help='dense MFMA first embedding layer instead of the sparse one')
    ap.add_argument('--no-roofline', action='store_true', help='skip the HIP-event roofline leg (used by the PMC child run)')
    ap.add_argument('--no-traffic', action='store_true', help='skip roofline.traffic (two short rocprofv3 PMC child runs at N=1)')
    ap.add_argument('--no-train', action='store_true', help='skip the "training" leg of the default run')
    ap.add_argument('--scenes', type=int, default=0, help='scenes per GPU instead of the configuration\'s (batch-size sweeps; the '
                    'headline stays on the configuration\'s own size)')
    ap.add_argument('--global-scenes', type=int, default=0, help='STRONG scaling: one fixed batch of this many scenes (BASELINE '
                    'config 3: --config directional --global-scenes 256) sharded over the ranks with parallel.shard_batch')
    ap.add_argument('--no-strong', action='store_true', help='skip the config-3 strong-scaling leg of the default run')
    ap.add_argument('--no-op-point', action='store_true', help='skip the trainer_default / per_scene_predict legs (batch_size 8 '
                    'training on fresh ragged batches, the evaluator\'s per-scene predictor call)')
    ap.add_argument('--no-sustain', action='store_true', help='skip the sustained-throughput leg (>= 2 s of back-to-back forwards)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # launched bare (`python bench.py --gpus N`): re-exec under torch.distributed.run, one rank per GPU, and hand its
        # exit status on -- the contract's launch line does the same from outside
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    distributed = world > 1
    # Test hooks for boxes with fewer GPUs than ranks (the N > 1 code path -- sharding, barrier, max over ranks, the
    # gradient all-reduce from inside the backward pass -- can then be exercised on ONE GPU): TNP_BENCH_SHARE_GPU=1 maps
    # rank r to device r % device_count, TNP_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    # Numbers measured that way mean nothing; the driver's runs use neither.
    if os.environ.get('TNP_BENCH_SHARE_GPU') == '1':
        local_rank %= max(torch.cuda.device_count(), 1)
    backend = os.environ.get('TNP_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = CONFIGS[args.config]
    if cfg.get('classical'):
        if distributed:
            raise SystemExit('--config classical is a single-GPU measurement')
        return bench_classical(args, cfg, device)
    model = build_model(cfg, device)
    is_sgan = bool(cfg.get('sgan'))
    if not is_sgan:
        model.kernel_variant = args.variant
        model.sparse_embedding = not args.dense
    if args.scenes > 0:
        cfg = dict(cfg, scenes=args.scenes)
    strong = args.global_scenes > 0
    if strong:
        # STRONG scaling (BASELINE config 3 is ONE batch of 256 scenes x 64 agents over 8 GPUs, assembled as one batch by
        # lstm/trainer.py:96-133): every rank builds the same global batch and keeps its shard of scenes; the shard
        # carries the batch-wide slot count and scene count (parallel.Shard), total work is fixed as N grows
        from trajnetplusplusbaselines_amd import parallel
        gxy, gsplit = synth.linear_crowd(args.global_scenes, cfg['agents'], seed=100)
        shard = parallel.shard_batch(gxy, torch.zeros(gxy.shape[1], 2), gsplit, rank, world)
        lo, hi = shard.track_range
        xy, split = gxy[:, lo:hi].contiguous(), shard.batch_split
        n_global, pad_to, scenes_local = shard.n_scenes_global, shard.pad_to, shard.n_scenes
        cfg = dict(cfg, scenes=scenes_local)
    else:
        # every rank gets its own shard of scenes (different seed): weak scaling, no data-path collective
        xy, split = synth.linear_crowd(cfg['scenes'], cfg['agents'], seed=100 + rank)
        n_global, pad_to, scenes_local = cfg['scenes'] * world, cfg['agents'], cfg['scenes']
    scenes_total = n_global
    M = xy.shape[1]
    observed = xy[:9].to(device)
    goals = torch.zeros(M, 2, device=device)

    if args.train and is_sgan:
        # one discriminator step + one generator step (g_steps = d_steps = 1, sgan/trainer.py:119-135, 258-300)
        from trajnetplusplusbaselines_amd.lstm import PredictionLoss
        from trajnetplusplusbaselines_amd.sgan.train_step import train_batch as sgan_train_batch
        g_opt = make_adam(model.generator.parameters())
        d_opt = make_adam(model.discriminator.parameters())
        criterion = PredictionLoss(keep_batch_dim=True)
        scene_dev = xy.to(device)

        def step():
            sgan_train_batch(model, g_opt, d_opt, criterion, scene_dev, goals, split, 'd')
            return sgan_train_batch(model, g_opt, d_opt, criterion, scene_dev, goals, split, 'g')
    elif args.train:
        from trajnetplusplusbaselines_amd.lstm import PredictionLoss
        from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
        optimizer = make_adam(model.parameters())   # lstm/trainer.py:497
        criterion = PredictionLoss()
        scene_dev = xy.to(device)

        def step():
            return train_batch(model, optimizer, criterion, scene_dev, goals, split, 9, 12, batch_size=n_global,
                               n_global_scenes=n_global, pad_to=pad_to, overlap=True)
    elif is_sgan:
        truth = xy[9:21].to(device)

        def step():
            return model(observed, goals, split, prediction_truth=truth, step_type='g', pred_length=12)
    else:
        def step():
            return model(observed, goals, split, n_predict=12, pad_to=pad_to)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.set_grad_enabled(args.train):
        # untimed preparation before the contract's W warm-up steps: half a second of the same step, so that code objects
        # are loaded, the caching allocator holds its blocks and the GPU has left its idle clocks (a 20-step timed region is
        # 28 ms: measured right after process start it once read 605 k instead of 1.0 M scene-steps/s)
        t_prime = time.perf_counter() + float(os.environ.get('TNP_BENCH_PRIME_S', '0.5'))
        while time.perf_counter() < t_prime:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sustained leg: the same forward back to back for >= 2.5 s (the timed region above is a few tens of ms at the
    #      driver's --steps 20: too short for a 5 s SMI sample to see the GPU busy, and a second, longer measurement of the
    #      same quantity is a cross-check of `value`) ----
    sustained = None
    if not args.train and not args.no_sustain and not args.no_roofline:
        n_sus = max(args.steps, int(2.5 / max(elapsed / args.steps, 1e-6)) + 1)     # `elapsed` is already the MAX over ranks
        with torch.no_grad():
            barrier()
            t0 = time.perf_counter()
            for _ in range(n_sus):
                step()
            barrier()
            t_s = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if distributed:
            dist.all_reduce(t_s, op=dist.ReduceOp.MAX)
        t_s = float(t_s.item())
        sustained = dict(value=scenes_total * 21 * n_sus / t_s, unit='scene-steps/s', forwards=n_sus, seconds=t_s,
                         ms_per_step=t_s / n_sus * 1e3)

    # ---- two batches in flight: the SAME forward on two HIP streams, alternating between two different 64-scene batches (what
    #      LSTMPredictor.predict_batches does for an evaluation set).  NOT the headline: `value` keeps one batch in flight. ----
    in_flight2 = None
    if not args.train and not is_sgan and not args.no_sustain and not args.no_roofline and not strong:
        xy_b, split_b = synth.linear_crowd(cfg['scenes'], cfg['agents'], seed=1000 + rank)
        obs_pair = [observed, xy_b[:9].to(device)]
        split_pair = [split, split_b]
        in_flight2 = dict(unit='scene-steps/s',
                          note='N independent batches of the headline shape in flight on N HIP streams (alternating between two crowds): '
                               'the kernels of one forward fill the per-kernel prologue / epilogue gaps of the others '
                               '(LSTMPredictor.predict_batches).  Reported beside the headline, which keeps ONE batch in flight.')
        for nfl in (2, 4):
            s_n = [torch.cuda.Stream(device=device) for _ in range(nfl)]
            n2 = max(args.steps, 200)
            with torch.no_grad():
                for i in range(2 * nfl):
                    with torch.cuda.stream(s_n[i % nfl]):
                        model(obs_pair[i & 1], goals, split_pair[i & 1], n_predict=12, pad_to=pad_to)
                barrier()
                t0 = time.perf_counter()
                for i in range(n2):
                    with torch.cuda.stream(s_n[i % nfl]):
                        model(obs_pair[i & 1], goals, split_pair[i & 1], n_predict=12, pad_to=pad_to)
                barrier()
                t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if distributed:
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            t2 = float(t2.item())
            in_flight2['in_flight_%d' % nfl] = dict(value=scenes_total * 21 * n2 / t2, forwards=n2, ms_per_step=t2 / n2 * 1e3)
        in_flight2['value'] = in_flight2['in_flight_2']['value']

    # ---- training leg: the optimisation step of the same model on the same shard, all-reduce included at N > 1 ----
    training = None
    training_failed = False
    if not args.train and not is_sgan and not args.no_train and not args.no_roofline:
        try:
            from trajnetplusplusbaselines_amd import parallel
            from trajnetplusplusbaselines_amd.lstm import PredictionLoss
            from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
            # same seed on every rank: replicas start identical.  On the headline configuration the seed is the one of
            # tests/golden/train_full.npz, whose 'synth' batch IS rank 0's batch here (synth.linear_crowd(64, 32, seed=100)):
            # the first optimisation step's loss is then a number the REFERENCE produced (oracle/gen_golden_r4.py), and the
            # leg fails if this run does not reproduce it -- the timed code is the code that was checked
            pinned = (args.config == 'social' and not strong and args.scenes == 0 and not args.dense)
            tmodel = build_model(cfg, device, seed=TRAIN_PIN_SEED if pinned else 0)
            tmodel.kernel_variant = args.variant
            optimizer = make_adam(tmodel.parameters())   # lstm/trainer.py:497
            # gradient all-reduce: 'overlap' (default) = from inside the backward pass, each gradient as soon as it is enqueued
            # (parallel.GradReducer); 'buckets' = flat persistent buckets launched asynchronously after the backward pass
            ar_mode = os.environ.get('TNP_BENCH_ALLREDUCE', 'overlap')
            buckets = parallel.GradBuckets(tmodel.parameters()) if (distributed and ar_mode == 'buckets') else None
            criterion = PredictionLoss()
            scene_dev = xy.to(device)
            t_steps, t_warm = max(5, min(args.steps, 30)), 3

            def tstep():
                return train_batch(tmodel, optimizer, criterion, scene_dev, goals, split, 9, 12, batch_size=n_global,
                                   n_global_scenes=n_global, pad_to=pad_to, buckets=buckets,
                                   overlap=(ar_mode == 'overlap'))
            loss_first = tstep()
            for _ in range(t_warm - 1):
                tstep()
            barrier()
            t0 = time.perf_counter()
            for _ in range(t_steps):
                loss_last = tstep()
            barrier()
            t_el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if distributed:
                dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
            t_el = float(t_el.item())
            grad_bytes = sum(p.numel() * 4 for p in tmodel.parameters() if p.grad is not None)
            training = dict(value=scenes_total * 21 * t_steps / t_el, unit='scene-steps/s', steps=t_steps, warmup=t_warm,
                            ms_per_step=t_el / t_steps * 1e3, loss_last=loss_last, loss_first=loss_first,
                            roofline=leg_roofline(3 * 19 * step_flops(tmodel, M, cfg['agents']), t_el / t_steps,
                                                  'forward (sparse first layer: gather bound) + data and weight gradients = 3 x the forward\'s '
                                                  'FLOPs, this rank\'s shard'),
                            workload='Trainer.train_batch of the same model on the same shard: teacher-forced forward, NLL loss, '
                                     'backward, Adam (' + type(optimizer).__module__ + ')%s' % (' + SUM all-reduce of %.1f MB of fp32 gradients over RCCL (%s)' % (
                                         grad_bytes / 1e6, 'launched from inside the backward pass as each gradient is enqueued, largest first'
                                         if ar_mode == 'overlap' else 'flat buckets after the backward pass') if distributed else ''),
                            allreduce_bytes=grad_bytes if distributed else 0,
                            scaling=dict(n=world, mode='strong' if strong else 'weak', global_scenes=scenes_total,
                                         scenes_this_rank=scenes_local, ms_per_step=t_el / t_steps * 1e3,
                                         allreduce_bytes=grad_bytes if distributed else 0,
                                         overlap=(ar_mode if distributed else None)))
            if pinned and rank == 0:
                rel_err = abs(loss_first - TRAIN_PIN_LOSS) / TRAIN_PIN_LOSS
                training['first_step_check'] = dict(reference_loss=TRAIN_PIN_LOSS, rel_err=rel_err, tol=TRAIN_PIN_RTOL,
                                                    ok=bool(rel_err < TRAIN_PIN_RTOL),
                                                    source='tests/golden/train_full.npz synth_loss: the reference\'s Trainer.train_batch '
                                                           'loss on this batch and these weights (oracle/gen_golden_r4.py)')
                if not rel_err < TRAIN_PIN_RTOL:
                    raise AssertionError('first optimisation step: loss %.6f, the reference computed %.6f' % (loss_first, TRAIN_PIN_LOSS))
            del tmodel, optimizer, buckets
        except Exception as exc:   # the inference line above is already measured: report the failure, then exit non-zero
            training = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:300]))
            training_failed = True
    # a failed leg on ANY rank must be known to all of them before anybody enters another collective: agree through the
    # rendezvous store (TCP, independent of RCCL), not through the process group a failed collective may have broken
    any_failed = training_failed
    if distributed:
        try:
            store = dist.distributed_c10d._get_default_store()
            store.add('tnp_train_failed', int(training_failed))
            store.add('tnp_train_done', 1)
            t_wait = time.time()
            while int(store.add('tnp_train_done', 0)) < world and time.time() - t_wait < 120:
                time.sleep(0.05)
            any_failed = int(store.add('tnp_train_failed', 0)) > 0 or int(store.add('tnp_train_done', 0)) < world
        except Exception:
            any_failed = training_failed

    # ---- strong-scaling leg (BASELINE config 3: ONE batch of 256 scenes x 64 agents, D-LSTM directional, sharded over the
    #      ranks): the default line of every N carries it, so one N = 1, 2, 4, 8 sweep of this script yields the weak-scaling
    #      curve of config 2 (`value`), its training step with the gradient all-reduce (`training`) AND config 3's
    #      strong-scaling curve for inference and training (`strong_scaling_config3`) ----
    strong3 = None
    if (not strong and not args.train and not is_sgan and args.config == 'social' and not args.no_strong
            and not args.no_roofline and not any_failed):
        try:
            strong3 = strong_scaling_leg(args, device, rank, world, barrier)
        except Exception as exc:
            strong3 = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:300]))
    strong4 = classical5 = None
    if strong3 is not None and 'error' not in strong3:
        try:
            strong4 = sgan_strong_leg(args, device, rank, world, barrier)
        except Exception as exc:
            strong4 = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:300]))
        if world == 1:
            try:
                classical5 = classical_leg(device)
            except Exception as exc:
                classical5 = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:300]))
    op_point = None
    if world == 1 and strong3 is not None and not args.no_op_point:
        try:
            op_point = operating_point_legs(device)
        except Exception as exc:
            op_point = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:300]))

    with torch.set_grad_enabled(args.train):
        # ---- roofline leg: same region again with HIP events around every launch of the dominant kernel ----
        L = _lib.lib()
        roof = None
        if rank == 0 and not args.train and not is_sgan and not args.no_roofline:
            import ctypes

            def profiled_pass():
                _lib.check(L.tnp_profile_begin(0), 'tnp_profile_begin')
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / args.steps
                ms_, n_ = ctypes.c_double(0.0), ctypes.c_int(0)
                _lib.check(L.tnp_profile_read(ctypes.byref(ms_), ctypes.byref(n_)), 'tnp_profile_read')
                disp = int(L.tnp_profile_dispatch_timed())
                L.tnp_profile_end()
                return ms_, n_, disp, wall
            # The library creates its event pairs on first use (two hipEventCreate per launch): on a slow host that makes the
            # first profiled pass host-bound, the queue runs dry between launches, the clocks fall and every launch reads ~1.7x
            # its duration (56 us against rocprofv3's 33 on one box of round 6).  The first pass only fills the pool; a pass whose
            # wall clock still is not the timed region's is repeated (at most twice) and the fastest pass counts.
            profiled_pass()
            passes = [profiled_pass()]
            while passes[-1][3] > 1.1 * elapsed / args.steps and len(passes) < 3:
                passes.append(profiled_pass())
            ms, n, dispatch_timed, leg_wall = min(passes, key=lambda q: q[0].value / max(q[1].value, 1))
            K0 = cfg['n'] * cfg['n'] * model.pool.pooling_dim
            N0 = model.pool.embedding_layers()[0].weight.shape[0]
            dense_flops = 2.0 * M * N0 * K0                # dense Linear(C*n*n -> N0) on the grid
            sparse = bool(model.sparse_embedding and cfg['type_'] == 'social')
            hits = None
            if sparse:
                # gather formulation: at most A-1 occupied cells per ego, C values each (SURVEY 8d "sparse lower
                # bound for the first embedding layer"); the kernel runs on the fp32 VALU, whose peak equals the
                # fp32 MFMA peak on CDNA4 (157.3 TFLOP/s)
                flops = 2.0 * M * (cfg['agents'] - 1) * model.pool.pooling_dim * N0
                _, pred_once = step()
                hits = occupied_cells_per_step(observed, pred_once, split, cfg['n'], 0.6)
                kname = ('pool_embed_regacc_kernel (winner tile from the positions + pool.embedding.0 on it: '
                         '%d egos x <=%d occupied cells x %d values -> %d)' % (M, cfg['agents'] - 1,
                                                                              model.pool.pooling_dim, N0))
            else:
                flops = dense_flops
                kname = 'gemm_nt_* (pool.embedding.0: [%d,%d]x[%d,%d]^T on v_mfma_f32_32x32x2_f32)' % (M, K0, N0, K0)
            if n.value > 0 and ms.value > 0:
                avg_s = ms.value / n.value * 1e-3
                achieved = flops / avg_s / 1e12
                # what an event pair costs with NOTHING between the two records, on the same stream: the bracket around a launch
                # contains this much that is not the kernel (rocprofv3's kernel durations do not), so `avg_launch_us` reads
                # higher than the committed rocprofv3 mean of the same kernel by about this amount
                pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
                step()
                for e0, e1 in pairs:
                    e0.record(); e1.record()
                    step()                                                       # keep the stream busy as in the measured region
                torch.cuda.synchronize()
                empty_us = float(np.median([e0.elapsed_time(e1) for e0, e1 in pairs])) * 1e3
                roof = dict(bound=('valu-f32' if sparse else 'mfma'), achieved=achieved, peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                            frac=achieved / FP32_MFMA_PEAK_TFLOPS, traffic=None, kernel=kname,
                            engine=('fp32 VALU FMA (same 157.3 TFLOP/s peak as the fp32 matrix cores; a stream of independent '
                                    'v_pk_fma_f32 alone sustains 127-138 TFLOP/s on this part: tools/diag/fp32_rates.hip, '
                                    'profiles/round6_fp32_rates.txt)' if sparse else 'fp32 MFMA'),
                            formulation=('sparse gather (algorithmic FLOPs = 2*M*(A-1)*C*N1)' if sparse
                                         else 'dense GEMM (algorithmic FLOPs = 2*M*N1*C*n*n)'),
                            launches=n.value, avg_launch_us=avg_s * 1e6, flops_per_launch=flops,
                            dense_equivalent_tflops=dense_flops / avg_s / 1e12,
                            share_of_step=ms.value * 1e-3 / elapsed,
                            profiled_pass_ms_per_forward=leg_wall * 1e3, profiled_passes=len(passes),
                            event_pair_empty_us=empty_us, dispatch_timed_launches=dispatch_timed,
                            event_note=('avg_launch_us: the two HIP events of every launch travel IN its dispatch (hipExtLaunchKernelGGL '
                                        'start / stop events = begin / end timestamps of the kernel\'s own AQL packet, the quantity the '
                                        'committed rocprofv3 kernel trace reports).  Two hipEventRecord calls around a launch read '
                                        'event_pair_empty_us with NOTHING between them on this stream; rounds 1-3 bracketed the launch that '
                                        'way and read ~3 us more than rocprofv3.') if dispatch_timed == n.value else
                                       ('avg_launch_us is the span between two HIP events recorded around each launch; two events with '
                                        'nothing between them already read event_pair_empty_us on this stream (about half of that sits '
                                        'inside a bracketed launch), which is why the rocprofv3 mean under profiles/ is ~3 us lower'))
                if hits is not None:
                    # the (A-1)-cell bound is an upper bound of the work; this is the work that was actually there
                    mean_hits = float(np.mean(hits))
                    hit_flops = 2.0 * mean_hits * model.pool.pooling_dim * N0
                    roof.update(hits_per_launch=mean_hits, hits_per_ego=mean_hits / M,
                                hits_per_ego_first_last_step=[hits[0] / M, hits[-1] / M],
                                achieved_on_hits=hit_flops / avg_s / 1e12,
                                frac_on_hits=hit_flops / avg_s / 1e12 / FP32_MFMA_PEAK_TFLOPS)
                if not args.no_traffic and world == 1 and not under_profiler():
                    child = ['--config', args.config] + (['--dense'] if args.dense else []) + \
                        (['--variant', str(args.variant)] if args.variant else [])
                    kre = 'pool_embed_regacc|pool_embed_cellsplit|pool_embed_sparse_kernel' if sparse else 'gemm_nt_'
                    kt = rocprof_kernel_us(kre, child + ['--no-train', '--no-strong', '--no-sustain', '--no-op-point'])
                    if kt is not None:
                        # the profiler's own mean of the same kernel, from a child run of this command under rocprofv3
                        # --kernel-trace (what profiles/*_kernel_stats.md holds), beside the HIP-event mean above
                        roof['rocprof_avg_launch_us'] = kt['avg_launch_us']
                        roof['rocprof_launches'] = kt['launches']
                        roof['frac_by_rocprof'] = flops / (kt['avg_launch_us'] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS
                    t = pmc_traffic(kre, child)
                    if t is not None:
                        roof['traffic'] = t['bytes']
                        roof['traffic_detail'] = t
                        # compulsory bytes of the layer: weights + output + (sparse: positions and per-track values, the
                        # winner tile is built in LDS; dense: the grid)
                        roof['compulsory_bytes'] = float(N0 * K0 * 4 + M * N0 * 4 +
                                                         (M * 8 + M * model.pool.pooling_dim * 4 if sparse else M * K0 * 4))

    # ---- whole-step roofline: how far the DESIGN (four launches per recurrent step) is from the chip, not only its
    #      dominant kernel: algorithmic FLOPs of every kernel of a step over the measured step time ----
    step_roof = None
    if rank == 0 and not args.train and not is_sgan and model.pool is not None and hasattr(model.pool, 'embedding_layers'):
        C = model.pool.pooling_dim
        layers = model.pool.embedding_layers()
        dims = [layers[0].weight.shape[1]] + [l.weight.shape[0] for l in layers]
        sparse = bool(model.sparse_embedding and cfg['type_'] == 'social')
        k = {}
        k['first_embedding_layer'] = (2.0 * M * (cfg['agents'] - 1) * C * dims[1]) if sparse else 2.0 * M * dims[0] * dims[1]
        k['other_embedding_layers'] = sum(2.0 * M * dims[i] * dims[i + 1] for i in range(1, len(dims) - 1))
        k['lstm_gates'] = 2.0 * M * (model.encoder.weight_ih.shape[1] + model.hidden_dim) * 4 * model.hidden_dim
        k['track_prepare'] = 2.0 * M * model.hidden_dim * (5 + (C if cfg['type_'] == 'social' else 0)) + 2.0 * M * 2 * 62
        per_step = sum(k.values())
        t_step = elapsed / args.steps / 19.0                             # every rank runs its own shard concurrently
        step_roof = dict(flops_per_recurrent_step=per_step, gflop_by_kernel={n: v / 1e9 for n, v in k.items()},
                         us_per_recurrent_step=t_step * 1e6, achieved=per_step / t_step / 1e12, peak=FP32_MFMA_PEAK_TFLOPS,
                         unit='TFLOP/s', frac=per_step / t_step / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         note='sum of the algorithmic FLOPs of the launches of one recurrent step (first layer: %s) over the '
                              'measured time of a step, per GPU' % ('gather bound 2*M*(A-1)*C*N1' if sparse else 'dense GEMM'))
        if sparse and roof is not None and 'hits_per_launch' in roof:
            on_hits = per_step - k['first_embedding_layer'] + 2.0 * roof['hits_per_launch'] * C * dims[1]
            step_roof['frac_on_hits'] = on_hits / t_step / 1e12 / FP32_MFMA_PEAK_TFLOPS

    if args.train and is_sgan:
        workload_mode = 'S-GAN training: one discriminator step + one generator step (k=3, Adam)'
    elif args.train:
        workload_mode = 'training step (Trainer.train_batch: teacher-forced forward, NLL loss, backward, Adam)'
    elif is_sgan:
        workload_mode = 'SGAN.forward (k=3 generator samples, teacher-forced, + real/fake discriminator scores)'
    else:
        workload_mode = 'inference forward (LSTM.forward, n_predict=12)'
    if rank == 0:
        value = scenes_total * 21 * args.steps / elapsed
        out = {
            'metric': 'scene-steps/sec (9 obs + 12 pred)',
            'value': value,
            'unit': 'scene-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': ('%s, ONE batch of %d scenes x %d agents x (9 obs + 12 pred) sharded over %d GPU(s), %s' % (
                                        cfg['name'], scenes_total, cfg['agents'], world, workload_mode) if strong else
                                    '%s, %d scenes x %d agents x (9 obs + 12 pred) per GPU, %s' % (
                                        cfg['name'], cfg['scenes'], cfg['agents'], workload_mode)),
                       'scenes_per_gpu': cfg['scenes'], 'global_scenes': scenes_total, 'agents_per_scene': cfg['agents'],
                       'mode': 'training step (fwd + bwd + Adam%s)' % (' + gradient all-reduce' if world > 1 else '') if args.train else 'inference forward',
                       'recurrent_steps_per_forward': (3 * 19 + 2 * 20) if is_sgan else 19, 'parallelism': 'dp%d (scene sharding)' % world,
                       'kernel_variant': args.variant,
                       'first_embedding_layer': 'dense mfma' if args.dense else 'sparse gather (social) / dense mfma'},
            'recurrent_scene_steps_per_s': scenes_total * ((3 * 19 + 2 * 20) if is_sgan else 19) * args.steps / elapsed,
            'roofline': roof,
            'step_roofline': step_roof,
            'sustained': sustained,
            'batches_in_flight': in_flight2,
            'training': training,
            'strong_scaling_config3': strong3,
            'strong_scaling_config4_sgan': strong4,
            'classical_config5': classical5,
            'trainer_default': (op_point or {}).get('trainer_default', op_point),
            'per_scene_predict': (op_point or {}).get('per_scene_predict', op_point),
        }
        if world == 1 and not args.no_cpu_baseline and not is_sgan:
            out['cpu_baseline'] = cpu_baseline(cfg, xy, split)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if any_failed:
        # a failed leg is a failed run: non-zero exit status on every rank (the JSON line above still carries what was
        # measured and `training.error`); a failed collective leaves the group unusable, so nobody waits on it
        sys.stdout.flush()
        sys.stderr.write('bench.py: the training leg failed on at least one rank\n')
        os._exit(3)
    if distributed:
        dist.barrier()   # rank 0 ran the roofline / cpu_baseline legs; leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
