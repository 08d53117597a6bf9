# the test of tools/experiments/grid_linear.hip (append to tests/test_gpu_lstm.py with the plumbing patch applied)
@pytest.mark.parametrize('type_,n,arch,dims', [('directional', 12, 'one_layer', []), ('occupancy', 8, 'two_layer', [96]), ('occupancy', 4, 'one_layer', []),
                                               ('directional', 16, 'two_layer', [128])])
def test_grid_built_inside_the_first_layer_agrees_with_the_two_launch_form(type_, n, arch, dims):
    """Round 6, csrc/grid_linear.hip: occupancy / directional grids are built in the prologue of their first embedding layer (no
    grid in HBM, one launch less per step).  Against the grid kernel + GEMM form (tuning knob grid_linear = 0): same NaN pattern,
    2e-5 on every output -- ragged scenes with absent tracks (cell-0 clobber by padded slots), scenes larger than a wave, a
    batch whose 16-ego tiles straddle scenes, free-running and teacher-forced, training gradients."""
    from trajnetplusplusbaselines_amd import _lib
    from trajnetplusplusbaselines_amd.lstm import PredictionLoss
    torch.manual_seed(13)
    pool = GridBasedPooling(type_=type_, hidden_dim=128, cell_side=0.6, n=n, out_dim=64, embedding_arch=arch, layer_dims=dims)
    model = LSTM(pool=pool).cuda().eval()
    try:
        for xy, split in (synth.ragged_crowd(9, 1, 23, seed=2, nan_frac=0.25), synth.ragged_crowd(3, 60, 90, seed=3), synth.linear_crowd(4, 7, seed=4)):
            goals = torch.zeros(xy.shape[1], 2)
            outs = {}
            for fused in (0, 1):
                _lib.tuning_set('grid_linear', fused)
                with torch.no_grad():
                    outs[fused] = model(xy[:9], goals, split, n_predict=12) + model(xy[:9], goals, split, prediction_truth=xy[9:20].clone())
            for a, b in zip(outs[0], outs[1]):
                assert torch.equal(torch.isnan(a), torch.isnan(b))
                assert (torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max().item() < 2e-5
        model.train()
        xy, split = synth.ragged_crowd(6, 2, 30, seed=5, nan_frac=0.2)
        xyd, goals = xy.cuda(), torch.zeros(xy.shape[1], 2, device='cuda')
        grads = {}
        for fused in (0, 1):
            _lib.tuning_set('grid_linear', fused)
            model.zero_grad(set_to_none=True)
            rel, _ = model(xyd[:9], goals, split, prediction_truth=xyd[9:20])
            PredictionLoss()(rel[-12:], xyd[9:21] - xyd[8:20], split).backward()
            grads[fused] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        for ga, gb in zip(grads[0], grads[1]):
            assert (ga - gb).abs().max().item() <= 2e-5 * max(1.0, ga.abs().max().item())
    finally:
        _lib.tuning_set('grid_linear', 1)
