"""az_tower_forward_static -- the entry point that allocates nothing (activation images in the library's
per-GPU static array) -- against az_tower_forward_dev, through the C ABI, bit for bit: two calls in a row,
the second with a smaller batch (the array keeps the first call's rows)."""
import numpy as np
import pytest

from alphazero_amd import capi, synth

pytestmark = pytest.mark.gpu


def _both(eng, boards_t, B, torch):
    A = eng.A
    outs = []
    for fn in (eng.tower_forward_dev, eng.tower_forward_static):
        pol = torch.full((B, A), -7.0, dtype=torch.float32, device="cuda")
        val = torch.full((B, 2), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        fn(boards_t.data_ptr(), B, pol.data_ptr(), val.data_ptr())
        eng.sync()
        outs.append((pol.cpu().numpy(), val.cpu().numpy()))
    return outs


@pytest.mark.parametrize("N,Cs,prec", [(9, [256, 256, 256], capi.AZ_PREC_FP16), (19, synth.NETCONFIG2, capi.AZ_PREC_BF16)],
                         ids=["9x9-256ch-fp16", "19x19-128ch-bf16"])
def test_static_scratch_forward_matches_the_allocating_entry_point(N, Cs, prec):
    import torch
    B1, B2 = 41, 7                                        # ragged last tile; then a smaller call
    eng = capi.Engine(N, 5, B1, Cs=Cs, precision=prec)
    assert not eng.tower_is_fused()                       # the per-layer path is the one with scratch
    eng.load_weights(synth.make_blob(N, 3, Cs, 91, phead_gain=4.0))
    boards = synth.random_boards(B1, N, seed=91)
    bt = torch.from_numpy(boards).cuda()
    (p_ref, v_ref), (p_st, v_st) = _both(eng, bt, B1, torch)
    assert np.isfinite(p_ref).all() and np.allclose(p_ref.sum(1), 1.0, atol=1e-4)
    assert np.array_equal(p_st, p_ref) and np.array_equal(v_st, v_ref)
    # second call, smaller: different boards, so stale rows of the first call would show
    boards2 = synth.random_boards(B2, N, seed=92)
    bt2 = torch.from_numpy(boards2).cuda()
    (p_ref2, v_ref2), (p_st2, v_st2) = _both(eng, bt2, B2, torch)
    assert np.array_equal(p_st2, p_ref2) and np.array_equal(v_st2, v_ref2)
    assert not np.array_equal(p_ref2, p_ref[:B2])
    # the static array bounds the batch, not az_config.G: one board more than the context was built for
    # works, one board more than the array holds is refused
    cap = eng.tower_static_max_batch()
    assert cap > B1
    big = torch.from_numpy(synth.random_boards(B1 + 1, N, seed=93)).cuda()
    pol = torch.empty((B1 + 1, eng.A), dtype=torch.float32, device="cuda")
    val = torch.empty((B1 + 1, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.tower_forward_static(big.data_ptr(), B1 + 1, pol.data_ptr(), val.data_ptr())
    eng.sync()
    assert np.allclose(pol.cpu().numpy().sum(1), 1.0, atol=1e-4)
    with pytest.raises(capi.AzError, match="static scratch"):
        eng.tower_forward_static(big.data_ptr(), cap + 1, pol.data_ptr(), val.data_ptr())
    with pytest.raises(capi.AzError):
        eng.tower_forward_dev(big.data_ptr(), B1 + 1, pol.data_ptr(), val.data_ptr())   # the allocating one is bound by G


def test_static_entry_point_on_a_fused_context_is_the_fused_kernel():
    """A context whose tower is one fused launch has no scratch at all: the entry point runs that launch."""
    import torch
    N, Cs, B = 9, synth.NETCONFIG2, 12
    eng = capi.Engine(N, 5, B, Cs=Cs)
    assert eng.tower_is_fused()
    eng.load_weights(synth.make_blob(N, 3, Cs, 94))
    bt = torch.from_numpy(synth.random_boards(B, N, seed=94)).cuda()
    (p_ref, v_ref), (p_st, v_st) = _both(eng, bt, B, torch)
    assert np.array_equal(p_st, p_ref) and np.array_equal(v_st, v_ref)
