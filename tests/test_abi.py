"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, the host classes mirror the reference's constructor / state_dict contract, and the
product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests import helpers
from trajnetplusplusbaselines_amd import _lib
from trajnetplusplusbaselines_amd import data as trajdata
from trajnetplusplusbaselines_amd.lstm import LSTM, GridBasedPooling, LSTMPredictor, drop_distant
from trajnetplusplusbaselines_amd.lstm.modules import InputEmbedding


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.exported_symbols_in_header()
    assert len(names) >= 12
    for name in names:
        assert hasattr(L, name), name
    assert L.tnp_abi_version() == _lib.ABI_VERSION == 7   # 2: tnp_lstm_model gained Wx / bx; 3: tnp_step_saves gained winners; 4: scene_slots; 5: Wp0_quad_major; 6: tnp_pool_pair_cells_autograd, tnp_bwd_sweep.cellwin_all; 7: tnp_lstm_model.pool_size / blur_size, evaluator-feed entry points


def test_struct_layout_matches_header_size():
    # 13 int32 + 4 float + 23 pointers + 1 int32 (+ padding) -- guards against drift between header and ctypes
    assert ctypes.sizeof(_lib.LstmModel) == 13 * 4 + 4 * 4 + 4 + 23 * 8 + 8 + 6 * 8 + 8 + 2 * 4


@pytest.mark.parametrize('kind', ['vanilla', 'occupancy', 'directional', 'social', 'social_goals'])
def test_state_dict_contract_matches_reference(kind):
    sd, cfg, _ = helpers.load_lstm_case(kind)
    model = helpers.build_amd_model(sd, cfg, device='cpu')
    ours = model.state_dict()
    assert list(ours.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(ours[k].shape) == tuple(sd[k].shape), k
        assert np.array_equal(ours[k].numpy(), sd[k]), k


def test_no_cpu_fallback():
    model = LSTM(pool=GridBasedPooling(type_='occupancy', n=4, out_dim=16))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(torch.zeros(9, 4, 2), torch.zeros(4, 2), torch.tensor([0, 4]), n_predict=12)
    with pytest.raises(AssertionError):   # reference lstm/lstm.py:197 XOR contract
        model(torch.zeros(9, 4, 2), torch.zeros(4, 2), torch.tensor([0, 4]))


def test_unsupported_options_raise():
    with pytest.raises(NotImplementedError):
        GridBasedPooling(type_='dir_social')
    with pytest.raises(NotImplementedError):
        GridBasedPooling(pretrained_pool_encoder=torch.nn.Sequential(torch.nn.Linear(4, 4)))
    with pytest.raises(ValueError):   # pool_to_input=False adds the pooled vector to the hidden state: out_dim == H
        LSTM(pool=GridBasedPooling(out_dim=32), hidden_dim=128, pool_to_input=False)
    # 'lstm_layer' behaves like 'one_layer' in the reference's forward and keeps its extra parameters
    assert 'pool_lstm.weight_ih' in GridBasedPooling(embedding_arch='lstm_layer', out_dim=16).state_dict()


def test_start_tags():
    # reference tests/test_lstm_modules.py:5-14
    emb = InputEmbedding(2, 4, 1.0)
    assert emb.start_enc(torch.zeros(1, 2)).numpy().tolist() == [[0.0, 0.0, 1.0, 0.0]]
    assert emb.start_dec(torch.zeros(1, 2)).numpy().tolist() == [[0.0, 0.0, 0.0, 1.0]]


def test_drop_distant():
    # reference tests/test_lstm_loss.py:46-60 adapted
    xy = np.array([[[0.0, 0.0], [1.0, 1.0], [20.0, 20.0], [np.nan, np.nan]],
                   [[0.0, 1.0], [1.5, 1.0], [20.0, 21.0], [3.0, 1.0]]])
    kept, mask = drop_distant(xy, r=6.0)
    assert mask.tolist() == [True, True, False, True]
    assert kept.shape == (2, 3, 2)


def test_paths_to_xy_and_center_scene_roundtrip():
    rows = [[trajdata.TrackRow(f, 7, float(f), 2.0 * f) for f in range(0, 50, 10)],
            [trajdata.TrackRow(f, 9, -float(f), 1.0) for f in range(10, 40, 10)]]
    xy = trajdata.paths_to_xy(rows)
    assert xy.shape == (5, 2, 2)
    assert np.isnan(xy[0, 1]).all() and np.isnan(xy[4, 1]).all()
    assert xy[2, 1].tolist() == [-20.0, 1.0]
    full = np.random.RandomState(0).randn(12, 3, 2)
    c, rot, center = trajdata.center_scene(full, obs_length=9)
    assert np.allclose(c[8, 0], 0.0)
    d = c[8, 0] - c[7, 0]
    assert abs(d[0]) < 1e-12 and d[1] > 0
    assert np.allclose(trajdata.inverse_scene(c, rot, center), full)


def test_ctypes_mirrors_have_the_c_struct_sizes():
    """The ctypes Structures of the host layer mirror the C structs of include/trajnet_hip.h field by field; a field added
    on one side only changes the size (or shifts every later pointer), which this catches without a GPU."""
    import ctypes
    from trajnetplusplusbaselines_amd.lstm import training
    L = _lib.lib()
    from trajnetplusplusbaselines_amd import optim
    mirrors = [_lib.LstmModel, _lib.LstmExtras, training.StepSaves, training.TrainSaves, training.BwdSweep, training.WgradProblem,
               optim.AdamTensor, training.TransposeProblem]
    for which, cls in enumerate(mirrors):
        assert L.tnp_abi_sizeof(which) == ctypes.sizeof(cls), cls.__name__
    assert L.tnp_abi_sizeof(99) == 0


def test_scene_replication_and_pool_kinds():
    """host helpers of the batched S-GAN samples and of the training sweep (no GPU needed)"""
    from trajnetplusplusbaselines_amd.sgan.sgan import _replicate_scenes, _scene_local
    from trajnetplusplusbaselines_amd.lstm.training import _pool_kind
    from trajnetplusplusbaselines_amd.lstm import (NearestNeighborMLP, HiddenStateMLPPooling, AttentionMLPPooling,
                                                   NearestNeighborLSTM, TrajectronPooling)
    split = torch.tensor([0, 3, 4, 9])
    assert _replicate_scenes(split, 1).tolist() == [0, 3, 4, 9]
    assert _replicate_scenes(split, 3).tolist() == [0, 3, 4, 9, 12, 13, 18, 21, 22, 27]
    assert _replicate_scenes([0, 2], 2).tolist() == [0, 2, 4]
    grid = GridBasedPooling(type_='directional', n=4, out_dim=16)
    assert _scene_local(None) and _scene_local(grid) and not _scene_local(TrajectronPooling(hidden_dim=32, out_dim=8))
    kinds = {None: 'none', grid: 'grid', NearestNeighborMLP(n=2, out_dim=8): 'nn',
             HiddenStateMLPPooling(hidden_dim=32, mlp_dim=48, mlp_dim_spatial=16, mlp_dim_vel=16, out_dim=8): 'hiddenmlp',
             AttentionMLPPooling(hidden_dim=32, mlp_dim=48, mlp_dim_spatial=16, mlp_dim_vel=16, out_dim=8): 'attention',
             NearestNeighborLSTM(n=2, hidden_dim=32, out_dim=8): 'stateful',
             TrajectronPooling(hidden_dim=32, out_dim=8): 'stateful'}
    for pool, kind in kinds.items():
        assert _pool_kind(pool) == kind


def test_header_is_plain_c_and_links(tmp_path):
    """include/trajnet_hip.h compiles as C99, a C program links against libtrajnet_hip.so and the host-only entry points
    answer without a GPU (tests/c/abi_smoke.c)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'trajnetplusplusbaselines_amd', 'lib')
    _lib.lib()                                      # builds the library if needed
    exe = str(tmp_path / 'abi_smoke')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(root, 'include'),
                    os.path.join(root, 'tests', 'c', 'abi_smoke.c'), '-o', exe, '-L', libdir, '-ltrajnet_hip',
                    '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith('ok abi')
