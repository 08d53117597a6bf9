"""hipGraph replays of a vanilla LSTM forward (the shortest kernels), copy in / replay / clone on the default stream, on a side
stream and through _GraphedForward as shipped.  History: while the sequence driver still used hipMemsetAsync (a memset NODE
in the captured graph) the default-stream variant returned, from the third replay on, outputs that belonged to no input, and a
device synchronisation anywhere in the call -- or a side stream -- hid it; tools/diag/graph_memset_probe.py shows the cause
(memset nodes lose bytes on replay), and with kernel fills every variant is right.  usage (gpurun): python tools/diag/graph_race_probe.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from trajnetplusplusbaselines_amd import synth                      # noqa: E402
from trajnetplusplusbaselines_amd import lstm as L                  # noqa: E402

torch.manual_seed(3)
model = L.LSTM(pool=None).cuda().eval()
crowds = []
for seed in range(12):
    xy, split = synth.ragged_crowd(5, 3, 40, seed=70)
    xy = xy + 0.3 * torch.randn(xy.shape, generator=torch.Generator().manual_seed(seed))
    crowds.append((xy, split))
M = crowds[0][0].shape[1]
goals = torch.zeros(M, 2)
with torch.no_grad():
    eager = [model(xy[:9], goals, split, n_predict=12)[1].cpu().numpy() for xy, split in crowds]
    from trajnetplusplusbaselines_amd.lstm import lstm as LL
    def make_call(sync_before, sync_after):
        def call(self, observed, goals, truth):
            self.obs.copy_(observed, non_blocking=True)
            if self.goals is not None:
                self.goals.copy_(goals, non_blocking=True)
            if sync_before:
                torch.cuda.synchronize()
            self.graph.replay()
            if sync_after:
                torch.cuda.synchronize()
            self.replays += 1
            return self.rel.clone(), self.pred.clone()
        return call
    for variant in ('product', 'bare_default_stream', 'bare_side_stream'):
        if variant != 'product':
            LL._GraphedForward.__call__ = make_call(False, False)
        model._graphs = None
        res = []
        side = torch.cuda.Stream()
        if variant == 'bare_side_stream':
            torch.cuda.set_stream(side)
        else:
            torch.cuda.set_stream(torch.cuda.default_stream())
        for i, (xy, split) in enumerate(crowds):
            obs = xy[:9].cuda() if variant == 'device_inputs' else xy[:9]
            if variant == 'sync_before_replay' or variant == 'sync_after_replay':
                # monkeypatch through the entry
                pass
            got = model(obs, goals, split, n_predict=12, graph=True)[1]
            got = got.cpu().numpy()
            same = [j for j in range(len(crowds)) if np.array_equal(got, eager[j], equal_nan=True)]
            res.append((i, same))
        print(variant, res)
