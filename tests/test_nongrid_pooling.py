"""Non-grid interaction modules (reference lstm/non_gridbased_pooling.py: NearestNeighborMLP, HiddenStateMLPPooling):
oracle and HIP path against the reference's own outputs (tests/golden/nongrid_cases.npz)."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

GOLD = np.load(os.path.join(helpers.GOLDEN, 'nongrid_cases.npz'))
KINDS = ['nn', 'hiddenstatemlp']


def state_dict(kind):
    pre = kind + '_sd_'
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def assert_rel_close(got, want, tol, what):
    """|got - want| <= tol * (1 + |want|): rows of absent egos pool the -100 fill values and are O(100)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape and (np.isnan(got) == np.isnan(want)).all(), what
    ok = ~np.isnan(want)
    err = (np.abs(got[ok] - want[ok]) / (1.0 + np.abs(want[ok]))).max()
    assert err <= tol, '%s: max scaled err %.3e > %.1e' % (what, err, tol)


def oracle_model(kind):
    return oracle.OracleModel(state_dict(kind), pool_type=kind, n=4)


@pytest.mark.parametrize('kind', KINDS)
def test_oracle_module_matches_reference(kind):
    om = oracle_model(kind)
    pre = kind + '_m_'
    got = oracle.pool_module(om, GOLD[pre + 'hidden'], GOLD[pre + 'obs1'], GOLD[pre + 'obs2'])
    assert_rel_close(got, GOLD[pre + 'out'], 5e-5, 'module output')   # rows of absent egos: 64 fill terms of -100 * w cancel
    got3 = oracle.pool_module(om, GOLD[pre + 'hidden'][:1, :3], GOLD[pre + 'obs1'][:1, :3], GOLD[pre + 'obs2'][:1, :3])
    assert_rel_close(got3, GOLD[kind + '_m3_out'], 5e-5, '3-slot scene')


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('batch', ['lin', 'rag'])
def test_oracle_lstm_forward_matches_reference(kind, batch):
    om = oracle_model(kind)
    pre = '%s_%s_' % (kind, batch)
    xy, split = GOLD[pre + 'xy'], GOLD[pre + 'split']
    goals = np.zeros((xy.shape[1], 2), np.float32)
    rel, pred = om.forward(xy[:9], goals, split, n_predict=12)
    helpers.assert_close_nan(rel, GOLD[pre + 'rel_npredict'], 2e-5, 'rel n_predict')
    helpers.assert_close_nan(pred, GOLD[pre + 'pred_npredict'], 2e-5, 'pred n_predict')
    rel, pred = om.forward(xy[:9], goals, split, prediction_truth=xy[9:20])
    helpers.assert_close_nan(rel, GOLD[pre + 'rel_truth'], 2e-5, 'rel truth')
    helpers.assert_close_nan(pred, GOLD[pre + 'pred_truth'], 2e-5, 'pred truth')
