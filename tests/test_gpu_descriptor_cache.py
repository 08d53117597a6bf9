"""LSTM._descriptor keeps the filled tnp_lstm_model between calls (lstm/lstm.py): what must invalidate it does, what must
not does not, and the value-dependent copies of the first embedding layer follow the parameter through optimiser steps."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def build(type_='social'):
    from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling
    torch.manual_seed(11)
    pool = GridBasedPooling(type_=type_, hidden_dim=128, cell_side=0.6, n=8, out_dim=64, embedding_arch='two_layer',
                            layer_dims=[128], latent_dim=16)
    return LSTM(pool=pool).cuda()


def raw(m):
    return ctypes.string_at(ctypes.addressof(m), ctypes.sizeof(m))


def test_second_call_returns_an_equal_copy_and_edits_do_not_stick():
    model = build()
    m1, keep1, _ = model._descriptor()
    assert model.__dict__['_desc_cache'] is not None
    m2, keep2, _ = model._descriptor()
    assert m1 is not m2 and raw(m1) == raw(m2)
    m2.Wn, m2.bn = None, None                     # what the backward does for a model without an output head
    m3, _, _ = model._descriptor()
    assert raw(m3) == raw(m1)


def test_a_moved_parameter_a_replaced_module_and_a_changed_setting_rebuild():
    model = build()
    m1, _, _ = model._descriptor()
    model.encoder.weight_ih.data = model.encoder.weight_ih.data.clone()
    m2, _, _ = model._descriptor()
    assert m2.enc_Wih != m1.enc_Wih and m2.enc_Wih == model.encoder.weight_ih.data_ptr()
    lin = model.pool.embedding[2]
    new = torch.nn.Linear(lin.in_features, lin.out_features).cuda()
    model.pool.embedding[2] = new
    m3, _, _ = model._descriptor()
    assert m3.Wp[1] == new.weight.data_ptr()
    model.kernel_variant = 262144
    m4, _, _ = model._descriptor()
    assert m4.variant == 262144
    model.sparse_embedding = False
    m5, _, _ = model._descriptor()
    assert not m5.Wp0_cell_major


def test_relaid_first_layer_follows_the_optimiser():
    """forward -> Adam step -> forward: the second forward must see the updated first-layer weight (its cell- / quad-major
    copies are refreshed although the descriptor itself is served from the cache)"""
    from trajnetplusplusbaselines_amd import synth
    model = build().eval()
    xy, split = synth.ragged_crowd(6, 2, 9, seed=3)
    xy = xy.cuda()
    goals = torch.zeros(xy.shape[1], 2, device='cuda')
    with torch.no_grad():
        a = model(xy[:9], goals, split, n_predict=12)[1]
        w = model.pool.embedding[0].weight
        w.mul_(1.5)                                # in place: same address, new version
        b = model(xy[:9], goals, split, n_predict=12)[1]
        assert model.__dict__['_desc_cache'] is not None
        fresh = build().eval()
        fresh.load_state_dict(model.state_dict())
        c = fresh(xy[:9], goals, split, n_predict=12)[1]
    prim = split[:-1]
    assert not torch.equal(a[:, prim], b[:, prim])
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(c))


def test_attention_pooling_is_never_served_from_the_cache():
    from trajnetplusplusbaselines_amd.lstm import LSTM
    from trajnetplusplusbaselines_amd.lstm.non_gridbased_pooling import AttentionMLPPooling
    model = LSTM(pool=AttentionMLPPooling(hidden_dim=128, out_dim=64)).cuda()
    model._descriptor()
    assert model.__dict__.get('_desc_cache') is None


def test_parameter_lists_follow_the_module_tree():
    model = build()
    names, params = model._named_parameter_lists()
    assert names == [n for n, _ in model.named_parameters()] and all(a is b for a, b in zip(params, model.parameters()))
    assert model._named_parameter_lists()[1] is params                       # served from the cache
    lin = model.pool.embedding[2]
    model.pool.embedding[2] = torch.nn.Linear(lin.in_features, lin.out_features).cuda()
    params2 = model._named_parameter_lists()[1]
    assert all(a is b for a, b in zip(params2, model.parameters())) and len(params2) == len(list(model.parameters()))
    model.encoder.register_parameter('extra', torch.nn.Parameter(torch.zeros(3, device='cuda')))
    names3, params3 = model._named_parameter_lists()
    assert 'encoder.extra' in names3 and len(params3) == len(params2) + 1
    model.encoder.weight_hh = torch.nn.Parameter(model.encoder.weight_hh.detach().clone())
    params4 = model._named_parameter_lists()[1]
    assert all(a is b for a, b in zip(params4, model.parameters()))
