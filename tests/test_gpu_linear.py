"""GPU parity of the fp32 MFMA GEMM (csrc/gemm_f32_mfma.hip) against the oracle's sequential-k fp32 Linear."""
import numpy as np
import pytest
import torch

from oracle import oracle
from trajnetplusplusbaselines_amd import _lib

pytestmark = pytest.mark.gpu

SHAPES = [
    (64, 64, 64), (128, 64, 32), (33, 70, 36), (1, 5, 128), (200, 130, 292), (257, 256, 288),
    (96, 62, 2), (31, 17, 9),          # K not a multiple of 4 -> scalar loader
    (2048, 256, 1024), (512, 1024, 4096),
    (2048, 448, 512),                  # 32x64 tiles need a second round of workgroups: automatic selection takes 32x128
    (2048, 256, 288), (300, 100, 96),  # K a multiple of 96, not of 64: split-K 3 of the pipelined kernel
    # round 5: shapes that select the eight-wave / small-batch tiles, with ragged edges in M and N
    (36, 256, 1024), (310, 250, 1024), (1000, 40, 512), (1, 33, 640), (2048, 448, 1024), (2050, 70, 512), (2048, 1024, 256),
]


@pytest.mark.parametrize('variant', [0, 12, 24, 25, 26, 27, 29, 30, 31, 33])   # automatic selection and the tile shapes it picks from; shapes
# the fast kernels cannot take (K not a multiple of the K step / of 4) fall through to the masked general kernel
@pytest.mark.parametrize('M,N,K', SHAPES)
def test_linear_matches_oracle(M, N, K, variant):
    rng = np.random.RandomState(M + 7 * N + 13 * K)
    x = rng.randn(M, K).astype(np.float32)
    x[rng.rand(M, K) < 0.5] = 0.0
    w = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32)
    for relu in (False, True):
        want = oracle.linear(x, w, b, relu=relu)
        got = _lib.linear_forward(torch.tensor(x).cuda(), torch.tensor(w).cuda(), torch.tensor(b).cuda(), relu=relu,
                                  variant=variant).cpu().numpy()
        # fp32 products are exact fmas on both sides; only the summation order differs
        tol = 2e-6 * np.sqrt(K) * max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got - want).max())
        assert err <= tol, (err, tol)


SKINNY_SHAPES = [(1, 5, 16), (4, 256, 1024), (36, 256, 1024), (37, 250, 1024), (310, 1024, 256), (100, 576, 512), (17, 33, 288),
                 (512, 256, 1024), (50, 70, 2048), (20, 16, 48)]


@pytest.mark.parametrize('variant', [40, 41, 42, 43, 44, 45])
@pytest.mark.parametrize('M,N,K', SKINNY_SHAPES)
def test_skinny_linear_matches_oracle(M, N, K, variant):
    """Round 6, csrc/gemm_skinny.hip: 16-track tiles, operands straight into registers, K over the waves of a workgroup (what the
    step's dense layers run on up to 512 tracks): ragged M / N, K in chunks of 16, every (column tiles, K split) form."""
    rng = np.random.RandomState(M + 7 * N + 13 * K)
    x = rng.randn(M, K).astype(np.float32)
    x[rng.rand(M, K) < 0.5] = 0.0
    w = (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32)
    xd, wd, bd = torch.tensor(x).cuda(), torch.tensor(w).cuda(), torch.tensor(b).cuda()
    for relu in (False, True):
        want = oracle.linear(x, w, b, relu=relu)
        out = torch.full((M + 1, N + 3), -7.0, device='cuda')
        _lib.linear_forward(xd, wd, bd, relu=relu, variant=variant, out=out[:M, :N])
        got = out.cpu().numpy()
        assert np.all(got[M:] == -7.0) and np.all(got[:, N:] == -7.0)             # nothing written past the ragged edges
        tol = 2e-6 * np.sqrt(K) * max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got[:M, :N] - want).max())
        assert err <= tol, (err, tol)
    # a row's sum does not depend on how many rows the call holds (batch invariance inside the regime)
    if M > 1:
        one = _lib.linear_forward(xd[M // 2:M // 2 + 1].contiguous(), wd, bd, relu=True, variant=variant)
        assert torch.equal(one[0], out[M // 2, :N])


def test_skinny_linear_refuses_unaligned_k():
    x, w = torch.randn(8, 40, device='cuda'), torch.randn(8, 40, device='cuda')
    with pytest.raises(RuntimeError, match='chunks of 16'):
        _lib.linear_forward(x, w, None, variant=40)
    got = _lib.linear_forward(x, w, None)                                          # automatic: falls to the 32-track kernels
    assert (got - x @ w.t()).abs().max().item() < 1e-4


def test_linear_transpose_detecting():
    """A = I with an asymmetric W: catches swapped rows / columns in the accumulator mapping."""
    K = 64
    x = np.eye(K, dtype=np.float32)
    w = (np.arange(96 * K, dtype=np.float32).reshape(96, K) % 251) / 16.0
    got = _lib.linear_forward(torch.tensor(x).cuda(), torch.tensor(w).cuda(), None).cpu().numpy()
    assert np.array_equal(got, w.T)


def test_linear_strided_output_slice():
    """The last embedding layer writes into a column slice of the LSTM input buffer (ldc > N)."""
    rng = np.random.RandomState(0)
    x = rng.randn(70, 40).astype(np.float32)
    w = rng.randn(24, 40).astype(np.float32)
    buf = torch.full((70, 100), -7.0, device='cuda')
    _lib.linear_forward(torch.tensor(x).cuda(), torch.tensor(w).cuda(), None, out=buf[:, 64:88])
    got = buf.cpu().numpy()
    assert np.all(got[:, :64] == -7.0) and np.all(got[:, 88:] == -7.0)
    np.testing.assert_allclose(got[:, 64:88], x @ w.T, atol=2e-5)


def test_transpose_grouped_matches_torch():
    """tnp_transpose_grouped: several matrices in one launch (more than twelve: a second launch), ragged shapes, strided
    destination."""
    import ctypes
    from trajnetplusplusbaselines_amd.lstm.training import TransposeProblem
    rng = np.random.RandomState(4)
    shapes = [(512, 320), (512, 128), (5, 128), (64, 2), (1, 1), (130, 70), (256, 1024)] * 2
    srcs = [torch.tensor(rng.randn(r, c).astype(np.float32)).cuda() for r, c in shapes]
    outs = [torch.full((c, r + 3), -7.0, device='cuda') for r, c in shapes]
    table = (TransposeProblem * len(shapes))(*[
        TransposeProblem(a.data_ptr(), a.stride(0), a.shape[0], a.shape[1], o.data_ptr(), o.stride(0)) for a, o in zip(srcs, outs)])
    _lib.check(_lib.lib().tnp_transpose_grouped(table, len(shapes), _lib.stream_ptr()), 'tnp_transpose_grouped')
    for a, o in zip(srcs, outs):
        assert torch.equal(o[:, :a.shape[0]], a.t()) and bool((o[:, a.shape[0]:] == -7.0).all())


@pytest.mark.parametrize('N1,C,G', [(1024, 16, 12), (128, 8, 8), (100, 4, 3), (192, 6, 5), (64, 32, 16)])
def test_weight_layouts_match_the_permutes(N1, C, G):
    """tnp_pool_embed_weight_layouts: the cell-major and quad-major copies of the first embedding layer against the tensor
    expressions INTEGRATION.md gives for them (strided source, ragged shapes; no quad-major copy when the shape rules it out)."""
    ncell = G * G
    rng = np.random.RandomState(N1 + C)
    full = torch.tensor(rng.randn(N1, C * ncell + 8).astype(np.float32)).cuda()
    W = full[:, 4:4 + C * ncell]
    quad = N1 % 64 == 0 and C % 4 == 0
    cm = torch.full((ncell, C, N1), float('nan'), device='cuda')
    qm = torch.full((ncell, N1 // 64, C // 4, 64, 4), float('nan'), device='cuda') if quad else None
    _lib.check(_lib.lib().tnp_pool_embed_weight_layouts(_lib.ptr(W), W.stride(0), N1, C, ncell, _lib.ptr(cm), _lib.ptr(qm),
                                                        _lib.stream_ptr()), 'tnp_pool_embed_weight_layouts')
    Wc = W.contiguous()
    assert torch.equal(cm, Wc.view(N1, C, ncell).permute(2, 1, 0).contiguous())
    if quad:
        assert torch.equal(qm, Wc.view(N1 // 64, 64, C // 4, 4, ncell).permute(4, 0, 2, 1, 3).contiguous())
