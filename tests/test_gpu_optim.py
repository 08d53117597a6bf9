"""tnp_adam_step (csrc/optim.hip) behind trajnetplusplusbaselines_amd.optim.Adam against torch.optim.Adam -- the optimiser the
reference's trainers construct (lstm/trainer.py:497: Adam(lr, weight_decay=1e-4)): same trajectories, same state_dict."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1024, 4096), (512, 320), (512,), (5, 128), (5,), (62, 2), (7, 3), (1,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ps.append(torch.nn.Parameter(torch.randn(9, generator=g).cuda()))          # never receives a gradient
    return ps


@pytest.mark.parametrize('weight_decay', [0.0, 1e-4])
def test_matches_torch_adam_over_steps(weight_decay):
    from trajnetplusplusbaselines_amd.optim import Adam
    ours, ref = _params(3), _params(3)
    o1 = Adam(ours, lr=1e-3, weight_decay=weight_decay)
    o2 = torch.optim.Adam(ref, lr=1e-3, weight_decay=weight_decay)
    g = torch.Generator().manual_seed(11)
    for step in range(12):
        for a, b in zip(ours[:-1], ref[:-1]):
            grad = (torch.randn(a.shape, generator=g) * (10.0 ** (step % 4 - 2))).cuda()
            a.grad, b.grad = grad.clone(), grad.clone()
        o1.step()
        o2.step()
        if step == 5:                                                        # scheduler-style change of the learning rate
            for og in (o1, o2):
                og.param_groups[0]['lr'] = 3e-4
    for a, b in zip(ours, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
    assert torch.equal(ours[-1], ref[-1]) and ours[-1].grad is None        # no gradient: untouched, weight decay included
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert s1['state'].keys() == s2['state'].keys()
    for k in s1['state']:
        assert float(s1['state'][k]['step']) == float(s2['state'][k]['step']) == 12.0
        assert torch.allclose(s1['state'][k]['exp_avg'], s2['state'][k]['exp_avg'], rtol=1e-5, atol=1e-6)   # torch's lerp_ rounds through an fma
        assert torch.allclose(s1['state'][k]['exp_avg_sq'], s2['state'][k]['exp_avg_sq'], rtol=1e-5, atol=1e-9)


def test_state_dict_round_trip_with_torch_adam():
    """a checkpoint of torch.optim.Adam resumes in ours and the other way round"""
    from trajnetplusplusbaselines_amd.optim import Adam
    a, b = _params(5)[:3], _params(5)[:3]
    o_t, o_n = torch.optim.Adam(a, lr=1e-3, weight_decay=1e-4), Adam(b, lr=1e-3, weight_decay=1e-4)
    g = torch.Generator().manual_seed(2)
    grads = [[torch.randn(p.shape, generator=g).cuda() for p in a] for _ in range(4)]
    for gs in grads[:2]:
        for p, q, gr in zip(a, b, gs):
            p.grad, q.grad = gr.clone(), gr.clone()
        o_t.step(); o_n.step()
    # swap the states: each continues from the other's checkpoint
    st_t, st_n = o_t.state_dict(), o_n.state_dict()
    o_t2, o_n2 = torch.optim.Adam(a, lr=1e-3, weight_decay=1e-4), Adam(b, lr=1e-3, weight_decay=1e-4)
    o_t2.load_state_dict(st_n); o_n2.load_state_dict(st_t)
    for gs in grads[2:]:
        for p, q, gr in zip(a, b, gs):
            p.grad, q.grad = gr.clone(), gr.clone()
        o_t2.step(); o_n2.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)


def test_unaligned_and_odd_sizes():
    from trajnetplusplusbaselines_amd.optim import Adam
    base = torch.randn(4099).cuda()
    p = torch.nn.Parameter(base[1:4098].clone())                             # 4097 elements
    q = torch.nn.Parameter(p.detach().clone())
    grad = torch.randn(4097).cuda()
    p.grad, q.grad = grad.clone(), grad.clone()
    Adam([p], lr=1e-2).step()
    torch.optim.Adam([q], lr=1e-2).step()
    assert torch.allclose(p, q, rtol=2e-6, atol=1e-7)


def test_updated_tensors_are_marked_as_modified():
    """the kernel writes through raw pointers: Tensor._version (what autograd's saved-tensor checks and our re-laid-out
    weight copies are keyed on) must still move"""
    from trajnetplusplusbaselines_amd.optim import Adam
    ps = _params(9)[:3]
    opt = Adam(ps, lr=1e-3)
    before = [p._version for p in ps]
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    assert all(p._version > v for p, v in zip(ps, before))
    assert all(opt.state[p]['exp_avg']._version > 0 for p in ps)


def test_social_lstm_trains_like_with_torch_adam():
    """Three optimisation steps of Social-LSTM (sparse first layer: the forward runs on cell- / quad-major COPIES of the first
    embedding layer that are rebuilt when the parameter's version changes) with the native Adam and with torch.optim.Adam:
    same losses, same parameters.  A stale copy shows from the second step on."""
    import copy
    from trajnetplusplusbaselines_amd import synth
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, PredictionLoss
    from trajnetplusplusbaselines_amd.lstm.train_step import train_batch
    from trajnetplusplusbaselines_amd.optim import Adam
    torch.manual_seed(5)
    pool = GridBasedPooling(type_='social', hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=16)
    m1 = LSTM(pool=pool).cuda()
    m2 = copy.deepcopy(m1)
    o1, o2 = Adam(m1.parameters(), lr=2e-2, weight_decay=1e-4), torch.optim.Adam(m2.parameters(), lr=2e-2, weight_decay=1e-4)
    xy, split = synth.ragged_crowd(6, 3, 9, seed=12)
    xy = xy.cuda()
    goals = torch.zeros(xy.shape[1], 2, device='cuda')
    l1 = [train_batch(m1, o1, PredictionLoss(), xy, goals, split, 9, 12) for _ in range(3)]
    l2 = [train_batch(m2, o2, PredictionLoss(), xy, goals, split, 9, 12) for _ in range(3)]
    assert l1[0] == l2[0]
    assert abs(l1[1] - l2[1]) <= 2e-4 * max(1.0, abs(l2[1])) and abs(l1[2] - l2[2]) <= 5e-4 * max(1.0, abs(l2[2])), (l1, l2)
    assert abs(l2[1] - l2[0]) > 1e-2 * abs(l2[0])             # the step is large enough for a stale forward to show
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), (n, float((a - b).abs().max()))
