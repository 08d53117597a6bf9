"""ms per inference forward and per optimisation step for every --type of the reference trainer at the config-2 batch
(64 scenes x 32 agents x 9+12 frames) -- measurement tool: python tools/bench_types.py"""
import sys, time, json, torch
sys.path.insert(0, '.')
from trajnetplusplusbaselines_amd import synth
from trajnetplusplusbaselines_amd.lstm import (LSTM, GridBasedPooling, NearestNeighborMLP, HiddenStateMLPPooling,
                                               AttentionMLPPooling, NearestNeighborLSTM, TrajectronPooling, PredictionLoss)
from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
from trajnetplusplusbaselines_amd.optim import Adam

POOLS = {
    'vanilla': lambda: None,
    'occupancy': lambda: GridBasedPooling(type_='occupancy', hidden_dim=128, cell_side=0.6, n=12, out_dim=256),
    'directional': lambda: GridBasedPooling(type_='directional', hidden_dim=128, cell_side=0.6, n=12, out_dim=256),
    'social': lambda: GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=16, out_dim=256,
                                       embedding_arch='two_layer', layer_dims=[1024], latent_dim=16),
    'nn': lambda: NearestNeighborMLP(n=4, out_dim=32),
    'hiddenstatemlp': lambda: HiddenStateMLPPooling(hidden_dim=128, out_dim=128),
    'attentionmlp': lambda: AttentionMLPPooling(hidden_dim=128, out_dim=128),
    'nn_lstm': lambda: NearestNeighborLSTM(n=4, hidden_dim=256, out_dim=32),
    'traj_pool': lambda: TrajectronPooling(hidden_dim=256, out_dim=32),
}
dev = torch.device('cuda', 0)
xy, split = synth.linear_crowd(64, 32, seed=100)
scene, goals = xy.to(dev), torch.zeros(xy.shape[1], 2, device=dev)
rows = []
for name, mk in POOLS.items():
    torch.manual_seed(0)
    model = LSTM(pool=mk()).to(dev)
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(scene[:9], goals, split, n_predict=12)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            model(scene[:9], goals, split, n_predict=12)
        torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 20
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)          # the native one-launch update (optim.py)
    crit = PredictionLoss()
    for _ in range(3):
        train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loss = train_batch(model, opt, crit, scene, goals, split, 9, 12, batch_size=64)
    torch.cuda.synchronize(); trn = (time.perf_counter() - t0) / 10
    rows.append(dict(type=name, forward_ms=round(1e3 * fwd, 3), train_step_ms=round(1e3 * trn, 3),
                     infer_scene_steps_per_s=round(64 * 21 / fwd), train_scene_steps_per_s=round(64 * 21 / trn), loss=round(loss, 4)))
    print(json.dumps(rows[-1]), flush=True)
