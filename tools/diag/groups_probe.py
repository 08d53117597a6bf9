"""ONE headline batch (64 scenes x 32 agents) run as G scene groups on G HIP streams (scenes are independent,
reference lstm/lstm.py:243-250): do the groups' kernels fill each other's prologue / epilogue gaps, and are the outputs
bit-equal to the single-sequence forward?  Python-level prototype of LSTM.forward(groups=G)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling

torch.manual_seed(0)
pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256, embedding_arch='two_layer',
                        layer_dims=[1024], latent_dim=16)
m = LSTM(pool=pool).eval().cuda()
scenes, agents = int(os.environ.get('SCENES', 64)), 32
xy, split = synth.linear_crowd(scenes, agents, seed=100)
obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
streams = [torch.cuda.Stream() for _ in range(8)]


def parts_of(G):
    per = scenes // G
    out = []
    for g in range(G):
        lo, hi = int(split[g * per]), int(split[(g + 1) * per])
        out.append((obs[:, lo:hi].contiguous(), goals[lo:hi].contiguous(), (split[g * per:(g + 1) * per + 1] - lo).clone()))
    return out


def forward(G, parts):
    if G == 1:
        return m(obs, goals, split, n_predict=12)[1]
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    outs = []
    for g in range(G):
        st = streams[g]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            outs.append(m(*parts[g], n_predict=12, pad_to=agents)[1])
        e2 = torch.cuda.Event(); e2.record(st); cur.wait_event(e2)
    return torch.cat(outs, dim=1)


if os.environ.get('PROBE_TRACE'):
    G = int(os.environ['PROBE_TRACE'])
    parts = parts_of(G)
    with torch.no_grad():
        for _ in range(60):
            forward(G, parts)
        torch.cuda.synchronize()
    sys.exit(0)

with torch.no_grad():
    for G in (2, 4, 8):          # ONE sub-batch alone on the current stream: the floor of a small forward
        part = parts_of(G)[0]
        for _ in range(10):
            m(*part, n_predict=12, pad_to=agents)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            m(*part, n_predict=12, pad_to=agents)
        torch.cuda.synchronize()
        print('one %d-scene sub-batch alone: %.3f ms per forward' % (scenes // G, (time.perf_counter() - t0) / 100 * 1e3), flush=True)

with torch.no_grad():
    ref = forward(1, None).clone()
    for G in (1, 2, 4, 8, 1, 2, 4):
        parts = parts_of(G)
        for _ in range(20):
            out = forward(G, parts)
        torch.cuda.synchronize()
        same = torch.equal(torch.nan_to_num(out), torch.nan_to_num(ref))
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            forward(G, parts)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        print('G=%d: %.3f ms per forward (host enqueue %.3f ms), %.0f scene-steps/s, bit-equal to G=1: %s, max|d| %.2e'
              % (G, t * 1e3, t_host * 1e3, scenes * 21 / t, same, float((torch.nan_to_num(out) - torch.nan_to_num(ref)).abs().max())), flush=True)

# ---- the same with one HOST THREAD per group (ctypes releases the GIL inside the native sequence call, so the groups' launches
# are enqueued in parallel and the host is not the limit): what the device does with G concurrent half / quarter batches ----
from concurrent.futures import ThreadPoolExecutor
pool_ex = ThreadPoolExecutor(max_workers=8)


def run_group(g, part, ev):
    st = streams[g]
    with torch.no_grad(), torch.cuda.stream(st):
        st.wait_event(ev)
        out = m(*part, n_predict=12, pad_to=agents)[1]
        e2 = torch.cuda.Event(); e2.record(st)
    return out, e2


def forward_threads(G, parts):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    futs = [pool_ex.submit(run_group, g, parts[g], ev) for g in range(G)]
    outs = []
    for f in futs:
        o, e2 = f.result()
        cur.wait_event(e2)
        outs.append(o)
    return torch.cat(outs, dim=1)


with torch.no_grad():
    for G in (2, 4, 2, 4):
        parts = parts_of(G)
        for _ in range(20):
            out = forward_threads(G, parts)
        torch.cuda.synchronize()
        same = torch.equal(torch.nan_to_num(out), torch.nan_to_num(ref))
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            forward_threads(G, parts)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n
        print('threads G=%d: %.3f ms per forward (host %.3f ms), %.0f scene-steps/s, bit-equal: %s'
              % (G, t * 1e3, t_host * 1e3, scenes * 21 / t, same), flush=True)
