"""Autograd through the STAND-ALONE ``GridBasedPooling.forward`` ("outputs participate in autograd"; VERDICT r4
"missing 5"): called outside ``LSTM.forward`` with parameters / inputs that require grad it must hand back the reference's
gradients, not silently none.  Fixture ``tests/golden/pool_grad.npz`` = reference autograd of sum(out * R)
(oracle/gen_golden_r5.py:pool_grad; lstm/gridbased_pooling.py:94-110, 227-305), crowds with an absent agent, an out-of-range
neighbour (cell (0,0) clobber) and two neighbours in one cell."""
import ast
import os

import numpy as np
import pytest
import torch

from trajnetplusplusbaselines_amd.lstm import GridBasedPooling

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pool_grad.npz'))
CASES = ['social', 'social_one', 'directional', 'directional_const', 'occupancy']


@pytest.mark.parametrize('name', CASES)
def test_standalone_pool_gradients_match_reference_autograd(name):
    pre = name + '_'
    kw = ast.literal_eval(str(G[pre + 'kw']))
    pool = GridBasedPooling(**kw)
    pool.load_state_dict({k[len(pre) + 2:]: torch.tensor(G[k]) for k in G.files if k.startswith(pre + 'w_')})
    pool = pool.cuda()
    directional = name.startswith('directional')
    l1 = torch.tensor(G[pre + 'obs1'], device='cuda', requires_grad=directional)
    l2 = torch.tensor(G[pre + 'obs2'], device='cuda', requires_grad=directional)
    h = torch.tensor(G[pre + 'hidden'], device='cuda', requires_grad=True)
    R = torch.tensor(G[pre + 'R'], device='cuda')
    y = pool(h, l1, l2)
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[pre + 'out'], rtol=0, atol=2e-5)
    (y * R).sum().backward()
    checked = 0
    for k, p in pool.named_parameters():
        want = G[pre + 'g_' + k]
        if want.size == 0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, 'no gradient reached %s' % k
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=0, atol=3e-5 * scale, err_msg=k)
        checked += 1
    assert checked >= 2
    if pre + 'g_hidden' in G.files:
        assert h.grad is not None
        np.testing.assert_allclose(torch.nan_to_num(h.grad).cpu().numpy(), G[pre + 'g_hidden'], rtol=0, atol=3e-5)
    else:
        assert h.grad is None or float(torch.nan_to_num(h.grad).abs().max()) == 0.0
    if directional:
        np.testing.assert_allclose(torch.nan_to_num(l2.grad).cpu().numpy(), G[pre + 'g_obs2'], rtol=0, atol=3e-5)
        np.testing.assert_allclose(torch.nan_to_num(l1.grad).cpu().numpy(), G[pre + 'g_obs1'], rtol=0, atol=3e-5)


def test_no_grad_path_is_bit_identical_to_the_autograd_path():
    pre = 'social_'
    kw = ast.literal_eval(str(G[pre + 'kw']))
    pool = GridBasedPooling(**kw)
    pool.load_state_dict({k[len(pre) + 2:]: torch.tensor(G[k]) for k in G.files if k.startswith(pre + 'w_')})
    pool = pool.cuda()
    o1, o2, h = (torch.tensor(G[pre + k], device='cuda') for k in ('obs1', 'obs2', 'hidden'))
    with torch.no_grad():
        a = pool(h, o1, o2)
    b = pool(h, o1, o2)
    assert b.requires_grad and not a.requires_grad
    assert torch.equal(a, b.detach())
