"""az_tower_forward_static -- the entry point that allocates nothing (activation images in the library's
per-GPU static array) -- against az_tower_forward_dev, through the C ABI, bit for bit: two calls in a row,
the second with a smaller batch (the array keeps the first call's rows)."""
import numpy as np
import pytest

from alphazero_amd import capi, synth

pytestmark = pytest.mark.gpu


def _both(eng, boards_t, B, torch):
    A = eng.A * eng.c_out
    outs = []
    for fn in (eng.tower_forward_dev, eng.tower_forward_static):
        pol = torch.full((B, A), -7.0, dtype=torch.float32, device="cuda")
        val = torch.full((B, eng.value_dim), -7.0, dtype=torch.float32, device="cuda")
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


@pytest.mark.parametrize("kind", ["t256_32", "t128", "blokus-shape"])
def test_static_scratch_on_the_other_kernels_and_shapes(monkeypatch, kind):
    """The same bit-for-bit agreement on the 32x32x16 and 128-row convolution kernels and on the Blokus tower's
    shape (5 planes in, 91 policy channels, 4 values), two calls in a row, the second smaller."""
    import torch
    if kind == "t256_32":
        monkeypatch.setenv("AZ_T256_32", "1")
    if kind == "t128":
        monkeypatch.setenv("AZ_CONV_T128", "1")
    if kind == "blokus-shape":
        N, Cs, kw = 20, synth.NETCONFIG2, dict(c_in=5, c_out=91, value_dim=4)
        blob = synth.make_blob(N, 5, Cs, 95, phead_gain=3.0, c_out=91, value_dim=4)
        mk = lambda B, seed: synth.random_planes(B, 5, N, seed=seed)          # noqa: E731
    else:
        N, Cs, kw = 9, [256, 256, 256], {}
        blob = synth.make_blob(N, 3, Cs, 95, phead_gain=3.0)
        mk = lambda B, seed: synth.random_boards(B, N, seed=seed)             # noqa: E731
    B1, B2 = 9, 4
    eng = capi.Engine(N, 5, B1, Cs=Cs, max_budget=1, **kw)
    assert not eng.tower_is_fused()
    eng.load_weights(blob)
    for B, seed in ((B1, 96), (B2, 97)):
        bt = torch.from_numpy(mk(B, seed)).cuda()
        (p_ref, v_ref), (p_st, v_st) = _both(eng, bt, B, torch)
        assert np.allclose(p_ref.sum(1), 1.0, atol=1e-4)
        assert np.array_equal(p_st, p_ref) and np.array_equal(v_st, v_ref)


def test_static_scratch_is_one_per_gpu_and_says_so():
    """Two contexts on one GPU share the array: a call while the other context's call is still running is refused
    with AZ_E_STATE instead of overwriting its activations; once that context has drained the call goes through."""
    import torch
    N, Cs, B = 19, synth.NETCONFIG2, 120
    a = capi.Engine(N, 5, B, Cs=Cs)
    b = capi.Engine(N, 5, 4, Cs=Cs)
    blob = synth.make_blob(N, 3, Cs, 98)
    a.load_weights(blob)
    b.load_weights(blob)
    big = torch.from_numpy(synth.random_boards(B, N, seed=98)).cuda()
    small = torch.from_numpy(synth.random_boards(4, N, seed=99)).cuda()
    pa, va = torch.empty((B, a.A), device="cuda"), torch.empty((B, 2), device="cuda")
    pb, vb = torch.empty((4, a.A), device="cuda"), torch.empty((4, 2), device="cuda")
    torch.cuda.synchronize()
    refused = 0
    for _ in range(3):                                     # a 120-board forward takes milliseconds; the next call comes microseconds later
        a.tower_forward_static(big.data_ptr(), B, pa.data_ptr(), va.data_ptr())
        try:
            b.tower_forward_static(small.data_ptr(), 4, pb.data_ptr(), vb.data_ptr())
        except capi.AzError as e:
            assert "in use by another context" in str(e)
            refused += 1
        a.sync()
        b.sync()
    assert refused >= 1
    want = pa.cpu().numpy().copy()
    b.tower_forward_static(small.data_ptr(), 4, pb.data_ptr(), vb.data_ptr())   # a has drained: accepted
    b.sync()
    a.tower_forward_static(big.data_ptr(), B, pa.data_ptr(), va.data_ptr())
    a.sync()
    assert np.array_equal(pa.cpu().numpy(), want)
