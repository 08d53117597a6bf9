# gates variant 22 (gate split x K split 2, eight waves) against the default 21, with the second layer's eight-wave tile on (TNP_GEMM2_V30=1)
python - <<'P'
import torch, sys
sys.path.insert(0, '.')
import bench
from trajnetplusplusbaselines_amd import synth
m = bench.build_model(bench.CONFIGS['social'], torch.device('cuda', 0), seed=1).eval()
xy, split = synth.ragged_crowd(64, 8, 40, seed=5, nan_frac=0.2)
obs, goals = xy[:9].cuda(), torch.zeros(xy.shape[1], 2).cuda()
outs = {}
with torch.no_grad():
    for v in (21, 22, 20):
        m.kernel_variant = v << 8
        outs[v] = m(obs, goals, split, n_predict=12)[1].clone()
for v in (22, 20):
    d = (torch.nan_to_num(outs[v]) - torch.nan_to_num(outs[21])).abs().max().item()
    print('gates variant %d vs 21 on a ragged NaN crowd: max |d position| %.2e, NaN pattern equal: %s' % (v, d, bool((torch.isnan(outs[v]) == torch.isnan(outs[21])).all())))
P
for i in 1 2 3; do for v in 0 5632; do echo -n "variant=$v: "; TNP_GEMM2_V30=1 python bench.py --variant $v --steps 200 --warmup 10 --no-cpu-baseline --no-traffic --no-train --no-strong --no-sustain --no-op-point 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f scene-steps/s, %.4f ms per forward' % (d['value'], d['ms_per_step']))"; done; done
