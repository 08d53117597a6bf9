"""GPU parity of the interaction-grid kernel (csrc/pool_grid.hip) through the GridBasedPooling API and the
C ABI: bit-exact against the reference's golden grids (occupancy / directional; social to fp32 rounding of
the hidden encoding) and against the oracle's integer cell ids."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import helpers
from trajnetplusplusbaselines_amd import _lib
from trajnetplusplusbaselines_amd.lstm import GridBasedPooling

pytestmark = pytest.mark.gpu

GRID_CASES = helpers.load_grid_cases()
FAST = [r for r in GRID_CASES if r['n'] * r['pool_size'] <= 64]


def make_pool(rec):
    pool = GridBasedPooling(cell_side=rec['cell_side'], n=rec['n'], hidden_dim=8, type_=rec['type'],
                            pool_size=rec['pool_size'], blur_size=rec['blur_size'], front=bool(rec['front']),
                            constant=rec['constant'], latent_dim=4, out_dim=8)
    if rec['type'] == 'social':
        with torch.no_grad():
            pool.hidden_dim_encoding.weight.copy_(torch.tensor(rec['Wh']))
            pool.hidden_dim_encoding.bias.copy_(torch.tensor(rec['bh']))
    return pool.to('cuda')


@pytest.mark.parametrize('rec', FAST, ids=[r['name'] for r in FAST])
def test_grid_matches_reference_golden(rec):
    pool = make_pool(rec)
    o1 = torch.tensor(rec['obs1']).cuda()
    o2 = torch.tensor(rec['obs2']).cuda()
    if rec['type'] == 'occupancy':
        g = pool.occupancies(o1, o2)
    elif rec['type'] == 'directional':
        g = pool.directional(o1, o2)
    else:
        g = pool.social(torch.tensor(rec['hidden']).cuda(), o1, o2)
    g = g.detach().cpu().numpy()      # (social grids carry autograd through hidden_dim_encoding, as the reference's do)
    ref = rec['grid']
    assert g.shape == ref.shape
    if rec['type'] == 'social' or rec['blur_size'] != 1:
        np.testing.assert_allclose(g, ref, rtol=0, atol=2e-6)
    else:
        assert np.array_equal(g, ref), float(np.abs(g - ref).max())


@pytest.mark.parametrize('rec', [r for r in GRID_CASES if 'tag_grid' in r],
                         ids=[r['name'] for r in GRID_CASES if 'tag_grid' in r])
def test_tag_grid_and_winner_table(rec):
    """occupancy() with per-pair tag values reproduces the reference's tag grid bit for bit, and the int16
    winner table agrees with the oracle's integer cell ids (last writer wins, cell-0 clobber)."""
    pool = make_pool(dict(rec, type='occupancy'))
    o2 = torch.tensor(rec['obs2']).cuda()
    B, N = o2.shape[:2]
    tags = torch.arange(1, N, dtype=torch.float32).view(1, 1, N - 1, 1).repeat(B, N, 1, 1).cuda()
    g = pool.occupancy(o2, tags, past_obs=torch.tensor(rec['obs1']).cuda()).cpu().numpy()
    assert np.array_equal(g, rec['tag_grid'])
    _, winners = pool._winner_grid(None, o2, _lib.POOL_OCCUPANCY, want_grid=False, want_winners=True)
    winners = winners.cpu().numpy()
    oi, inr = oracle.cell_ids(rec['obs2'], rec['n'], rec['cell_side'], front=bool(rec['front']))
    n2 = rec['n'] ** 2
    want = np.full((B * N, n2), -1, dtype=np.int16)
    for r in range(B * N):
        i = r % N
        for jj in range(N - 1):
            j = jj if jj < i else jj + 1
            want[r, oi.reshape(B * N, N - 1)[r, jj]] = j if inr.reshape(B * N, N - 1)[r, jj] else -1
    assert np.array_equal(winners, want)


@pytest.mark.parametrize('type_,n,agents', [('occupancy', 16, 32), ('directional', 12, 64), ('social', 16, 32),
                                            ('social', 16, 97), ('occupancy', 8, 130)])
def test_grid_random_large_vs_oracle(type_, n, agents):
    rng = np.random.RandomState(n * agents)
    B = 5
    obs2 = (rng.rand(B, agents, 2).astype(np.float32) * 8 - 4)
    obs1 = obs2 - rng.randn(B, agents, 2).astype(np.float32) * 0.3
    obs2[rng.rand(B, agents) < 0.15] = np.nan
    obs1[rng.rand(B, agents) < 0.1] = np.nan
    enc = rng.randn(B, agents, 16).astype(np.float32) if type_ == 'social' else None
    want = oracle.grid(type_, obs1, obs2, enc, n=n, cell_side=0.6, C=None if type_ != 'social' else 16)
    pool = GridBasedPooling(cell_side=0.6, n=n, type_=type_, latent_dim=16, out_dim=8).cuda()
    tid = _lib.POOL_TYPES[type_]
    vals = torch.tensor(enc).cuda() if enc is not None else None
    got, _ = pool._winner_grid(torch.tensor(obs1).cuda(), torch.tensor(obs2).cuda(), tid, values=vals)
    got = got.cpu().numpy().reshape(want.shape)
    assert np.array_equal(got, want)


def test_ragged_scenes_padded_slot_clobber():
    """Flat ragged layout: scenes shorter than n_max get the reference's padded-slot cell-0 clobber."""
    rng = np.random.RandomState(3)
    sizes = [5, 2, 7, 1, 7]
    n_max, n = max(sizes), 4
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    M = int(starts[-1])
    flat = (rng.rand(M, 2).astype(np.float32) * 2 - 1) * 2.0
    flat[::3] = np.float32(-1.9)          # park agents in cell (0, 0) relative to others
    padded = np.full((len(sizes), n_max, 2), np.nan, dtype=np.float32)
    for s, ns in enumerate(sizes):
        padded[s, :ns] = flat[starts[s]:starts[s + 1]]
    want = oracle.grid('occupancy', padded, padded, n=n, cell_side=1.0).reshape(len(sizes), n_max, -1)
    dev = torch.device('cuda')
    grid = torch.empty(M, n * n, dtype=torch.float32, device=dev)
    o = torch.tensor(flat).cuda()
    st = torch.tensor(starts).cuda()
    _lib.check(_lib.lib().tnp_pool_grid_forward(_lib.POOL_OCCUPANCY, _lib.ptr(o), _lib.ptr(o), None, 0, _lib.ptr(st),
                                                len(sizes), n_max, None, n, 1, 1.0, n / 2, n / 2, 0.0, _lib.ptr(grid),
                                                n * n, None, _lib.stream_ptr()), 'grid')
    got = grid.cpu().numpy()
    for s, ns in enumerate(sizes):
        assert np.array_equal(got[starts[s]:starts[s + 1]], want[s, :ns]), s


def test_duplicate_cells_single_threaded_reference():
    """tests/golden/dup_cells.npz through the HIP grid kernel: duplicate cells keep the last neighbour in ascending j,
    as the single-threaded reference does (occupancy / directional bit-exact, social within fp32 rounding of the
    hidden encoding; a wrong winner is off by O(1))."""
    import os
    z = np.load(os.path.join(helpers.GOLDEN, 'dup_cells.npz'))
    o1, o2 = torch.tensor(z['obs1']).cuda(), torch.tensor(z['obs2']).cuda()
    for type_ in ('occupancy', 'directional', 'social'):
        pool = GridBasedPooling(type_=type_, hidden_dim=128, cell_side=0.6, n=12, out_dim=32, latent_dim=8).cuda()
        if type_ == 'occupancy':
            g = pool.occupancies(o1.clone(), o2.clone())
        elif type_ == 'directional':
            g = pool.directional(o1.clone(), o2.clone())
        else:
            with torch.no_grad():
                pool.hidden_dim_encoding.weight.copy_(torch.tensor(z['Wh']))
                pool.hidden_dim_encoding.bias.copy_(torch.tensor(z['bh']))
            g = pool.social(torch.tensor(z['hidden']).cuda(), o1.clone(), o2.clone())
        g = g.detach().cpu().numpy()
        if type_ == 'social':
            np.testing.assert_allclose(g, z['grid_social'], rtol=0, atol=2e-6)
        else:
            assert np.array_equal(g, z['grid_' + type_]), type_


# ---- the scatter's backward as autograd defines it (round 4) ---------------------------------------------------------------
def _scatter_case(z, k, drop_trailing_padding):
    """flat-layout inputs of fixture case k; with drop_trailing_padding the all-NaN trailing slots of a scene are not tracks
    but padded slots (row_padded = N), which must give the same tables for the remaining rows"""
    import torch
    pre = 's%d_' % k
    obs1, obs2 = z[pre + 'obs1'], z[pre + 'obs2']
    B, N = obs2.shape[:2]
    keep = []
    for b in range(B):
        ns = N
        if drop_trailing_padding:
            while ns > 1 and np.isnan(obs2[b, ns - 1]).any() and np.isnan(obs1[b, ns - 1]).any():
                ns -= 1
        keep.append(ns)
    rows = [(b, j) for b in range(B) for j in range(keep[b])]
    o1 = np.stack([obs1[b, j] for b, j in rows]).astype(np.float32)
    o2 = np.stack([obs2[b, j] for b, j in rows]).astype(np.float32)
    starts = np.concatenate([[0], np.cumsum(keep)])
    row_base = np.concatenate([np.full(keep[b], starts[b]) for b in range(B)]).astype(np.int32)
    row_count = np.concatenate([np.full(keep[b], keep[b]) for b in range(B)]).astype(np.int32)
    row_padded = np.full(len(rows), N, dtype=np.int32)
    dev = lambda a: torch.tensor(a).cuda()
    return rows, dev(o1), dev(o2), dev(row_base), dev(row_count), dev(row_padded), B, N


@pytest.mark.parametrize('drop', [False, True])
def test_pair_cells_autograd_tables_and_scatter_backward_match_reference(drop):
    """tnp_pool_pair_cells_autograd == the numpy statement of the rule (which tests/test_oracle_golden.py pins to the
    reference's autograd, exactly), and tnp_social_scatter_backward / tnp_directional_scatter_backward on those tables
    reproduce the reference's gradients (tests/golden/scatter_grad.npz): a genuine neighbour in cell (0, 0) that an
    out-of-range / absent / padded slot of higher index clobbers gets NO gradient (lp_pool2d's zero derivative at the
    constant 0), every other in-range neighbour gets its cell's, duplicates included."""
    import os
    import torch
    from trajnetplusplusbaselines_amd import _lib
    z = np.load(os.path.join(helpers.GOLDEN, 'scatter_grad.npz'))
    L = _lib.lib()
    for k in range(int(z['num_cases'])):
        pre = 's%d_' % k
        n, cs, const = int(z[pre + 'n']), float(z[pre + 'cell_side']), float(z[pre + 'constant'])
        rows, o1, o2, rb, rc, rp, B, N = _scatter_case(z, k, drop)
        M = len(rows)
        raw = torch.empty(M, N, dtype=torch.int32, device='cuda')
        cells = torch.empty(M, N, dtype=torch.int32, device='cuda')
        win = torch.empty(M, N, dtype=torch.int32, device='cuda')
        _lib.check(L.tnp_pool_pair_cells_autograd(_lib.ptr(o2), _lib.ptr(rb), _lib.ptr(rc), _lib.ptr(rp), N, M, N, n, cs, n / 2.0,
                                                  n / 2.0, const, _lib.ptr(raw), _lib.ptr(cells), _lib.ptr(win), _lib.stream_ptr()),
                   'pair_cells_autograd')
        cells_h, win_h = cells.cpu().numpy(), win.cpu().numpy()
        for m, (b, i) in enumerate(rows):
            want_c, want_w = helpers.pair_cells_autograd_numpy(z[pre + 'obs2'][b], n, cs, const)
            ns = int(rc[m])
            np.testing.assert_array_equal(cells_h[m, :ns], want_c[i, :ns], err_msg='case %d row %d' % (k, m))
            np.testing.assert_array_equal(win_h[m, :ns], want_w[i, :ns], err_msg='case %d row %d (winner)' % (k, m))
            assert (cells_h[m, ns:] == -1).all()
        # social-type scatter backward: denc[j] = sum over the egos of the reference's per-pair gradients
        C = z[pre + 'dvalues'].shape[-1]
        ridx = {bj: m for m, bj in enumerate(rows)}
        dgrid = np.stack([z[pre + 'dgrid'][b * N + i].reshape(-1) for b, i in rows]).astype(np.float32)
        want = np.zeros((M, C), dtype=np.float64)
        for (b, i) in rows:
            for j in range(N):
                if j != i and (b, j) in ridx:
                    want[ridx[(b, j)]] += z[pre + 'dvalues'][b, i, j - (j > i)]
        denc = torch.empty(M, C, device='cuda')
        dg = torch.tensor(dgrid).cuda()
        _lib.check(L.tnp_social_scatter_backward(_lib.ptr(dg), dg.shape[1], _lib.ptr(cells), _lib.ptr(rb), _lib.ptr(rc), M, N, C,
                                                 n * n, _lib.ptr(denc), _lib.stream_ptr()), 'social_scatter_backward')
        np.testing.assert_allclose(denc.cpu().numpy(), want, rtol=0, atol=2e-5)
        # directional: gradient with respect to the velocities = d/d obs2 = - d/d obs1 of the reference
        dgrid2 = np.stack([z[pre + 'dir_dgrid'][b * N + i].reshape(-1) for b, i in rows]).astype(np.float32)
        dg2 = torch.tensor(dgrid2).cuda()
        dvel = torch.empty(M, 2, device='cuda')
        _lib.check(L.tnp_directional_scatter_backward(_lib.ptr(dg2), dg2.shape[1], _lib.ptr(cells), _lib.ptr(win), _lib.ptr(rb),
                                                      _lib.ptr(rc), _lib.ptr(o1), _lib.ptr(o2), M, N, n * n, _lib.ptr(dvel),
                                                      _lib.stream_ptr()), 'directional_scatter_backward')
        want2 = np.stack([z[pre + 'dir_dobs2'][b, i] for b, i in rows])
        want1 = np.stack([z[pre + 'dir_dobs1'][b, i] for b, i in rows])
        np.testing.assert_allclose(dvel.cpu().numpy(), want2, rtol=0, atol=2e-5)
        np.testing.assert_allclose(-dvel.cpu().numpy(), want1, rtol=0, atol=2e-5)
