"""GPU parity of the tower (stem / MFMA conv / heads) against fixtures produced
by the unmodified reference network.hpp, and against a torch fp32 restatement
at sizes the fixtures do not cover.  Tolerance: 1e-3 abs on the softmax
policy and on the value (the bar BASELINE.json states)."""
import glob
import os

import numpy as np
import pytest

from alphazero_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOWER_FIX = sorted(glob.glob(os.path.join(GOLDEN, "tower_*.npz")))
TOL = 1e-3


def torch_reference(N, Cs, blob, boards, c_out=1, value_dim=2):
    import torch
    import torch.nn.functional as F
    w = {k: torch.from_numpy(v.copy()) for k, v in synth.unpack_blob(blob, N, boards.shape[1], Cs, c_out, value_dim).items()}
    x = torch.from_numpy(boards.astype(np.float32))

    def bn(x, n):
        return F.batch_norm(x, w[n + ".running_mean"], w[n + ".running_var"], w[n + ".weight"], w[n + ".bias"], False, 0.0, 1e-5)
    x = F.relu(bn(F.conv2d(x, w["CBlock.conv.weight"], w["CBlock.conv.bias"], padding=1), "CBlock.bn"))
    for i in range(len(Cs) - 1):
        r = "RBlock_%d" % i
        h = F.relu(bn(F.conv2d(x, w[r + ".conv1.weight"], w[r + ".conv1.bias"], padding=1), r + ".bn1"))
        h = bn(F.conv2d(h, w[r + ".conv2.weight"], w[r + ".conv2.bias"], padding=1), r + ".bn2")
        x = F.relu(h + x)
    p = F.conv2d(x, w["PHead.conv.weight"], w["PHead.conv.bias"]).flatten(1).softmax(1)
    v = F.relu(bn(F.conv2d(x, w["VHead.conv.weight"], w["VHead.conv.bias"]), "VHead.bn")).flatten(1)
    v = F.relu(F.linear(v, w["VHead.fc1.weight"], w["VHead.fc1.bias"]))
    v = F.linear(v, w["VHead.fc2.weight"], w["VHead.fc2.bias"]).softmax(1) * 2 - 1
    return p.numpy(), v.numpy()


# Fixture cases that miss the one 1e-3 bound, per operand format, as measured on the MI355X (strict: a case that
# starts to pass, or a listed case that no longer exists, fails the suite).  fp16: the 41-convolution random-init
# tower with heads sharpened 20x reaches 1.02e-3 on 2 of its 256 boards (median 3.6e-4), and the towers with
# trained-network-like BatchNorm AND peaked heads amplify the operand rounding to 2e-3 .. 6e-3 (BatchNorm
# renormalises every layer, so a perturbation grows from block to block; the fp32 reference itself sits 20-40x
# further from an fp64 evaluation on them than on the random-init towers).  AZ_PREC_FP16X3 holds the bound on all.
FP16_MISSES = {"tower_n15_c256x21_sharp", "tower_n15_c256x21_spread_sharp", "tower_n15_cfg2_spread_sharp", "tower_n9_cfg2_spread_sharp"}
BF16_MISSES = FP16_MISSES | {"tower_n15_c256x21_spread", "tower_n15_cfg2_sharp", "tower_n15_cfg2_spread"}


def _fixture_cases():
    """Every reference fixture in the three operand formats against ONE tolerance."""
    cases = []
    for p in TOWER_FIX:
        stem = os.path.basename(p)[:-4]
        for prec, name, misses in ((capi.AZ_PREC_FP16, "fp16", FP16_MISSES), (capi.AZ_PREC_BF16, "bf16", BF16_MISSES),
                                   (capi.AZ_PREC_FP16X3, "fp16x3", set())):
            marks = [pytest.mark.xfail(reason="%s operands miss 1e-3 on this tower (see FP16_MISSES); AZ_PREC_FP16X3 is the parity mode" % name,
                                       strict=True)] if stem in misses else []
            cases.append(pytest.param(p, prec, id="%s-%s" % (os.path.basename(p), name), marks=marks))
    return cases


@pytest.mark.parametrize("path,prec", _fixture_cases())
def test_tower_matches_reference_fixture(path, prec):
    z = np.load(path)
    N, Cs, B = int(z["N"]), [int(c) for c in z["Cs"]], int(z["B"])
    c_in, c_out = (int(z["c_in"]), int(z["c_out"])) if "c_in" in z.files else (3, 1)
    blob = synth.fixture_blob(z)
    eng = capi.Engine(N, min(N, 5), B, Cs=Cs, precision=prec, c_in=c_in, c_out=c_out, max_budget=1)
    eng.load_weights(blob)
    pol, val = eng.tower_forward(z["boards"])
    perr, verr = np.abs(pol - z["policy"]).max(), np.abs(val - z["value"]).max()
    print(os.path.basename(path), "prec", prec, "policy err %.3g value err %.3g (max p %.3g)" % (perr, verr, z["policy"].max()))
    assert perr < TOL and verr < TOL
    assert eng.stats()["saturated"] == 0


def test_default_precision_is_fp16():
    """A zero-initialised az_config / a default Engine runs fp16 operands."""
    assert capi.AZ_PREC_FP16 == 0 and capi.AzConfig().precision == capi.AZ_PREC_FP16


@pytest.mark.parametrize("fused", [True, False])
def test_weights_outside_fp16_range_are_refused_or_fall_back(monkeypatch, fused):
    """BatchNorm statistics unrelated to the activations (running_var down to 1e-4 under unit-scale convolutions)
    fold into weights beyond +-65504, and a channel with a vanishing BatchNorm weight folds into a row of fp16
    subnormals: az_load_weights refuses both with AZ_E_ARG and keeps the weights in use, says what it found in
    az_weights_info, and with az_config.range_fallback takes the blob on bf16 operands instead."""
    if not fused:
        monkeypatch.setenv("AZ_NO_FUSED", "1")
    N, B, Cs = 9, 6, synth.NETCONFIG2
    good = synth.make_blob(N, 3, Cs, 77, phead_gain=5.0)
    boards = synth.random_boards(B, N, seed=77)
    lay, off = {}, 0
    for name, shape in synth.blob_layout(N, 3, Cs):
        lay[name] = (off, int(np.prod(shape)))
        off += int(np.prod(shape))
    over = good.copy()                                   # one channel: gamma / sqrt(var) = 3 / sqrt(1e-12 + 1e-5) ~ 950,
    o, _ = lay["RBlock_2.bn1.running_var"]; over[o + 7] = 1e-12      # times a convolution scaled up 4000x
    o, _ = lay["RBlock_2.bn1.weight"]; over[o + 7] = 3.0
    o, n = lay["RBlock_2.conv1.weight"]; over[o + 7 * 1152:o + 8 * 1152] *= 4000.0
    under = good.copy()                                  # one channel switched (almost) off: gamma = 1e-6
    o, _ = lay["RBlock_5.bn2.weight"]; under[o + 100] = 1e-6
    eng = capi.Engine(N, 5, B, Cs=Cs)
    assert eng.tower_is_fused() == fused
    eng.load_weights(good)
    info = eng.weights_info()
    assert info["overflow_weights"] == 0 and info["underflow_channels"] == 0 and info["precision_in_use"] == capi.AZ_PREC_FP16
    assert 0 < info["min_channel_max"] <= info["max_abs_weight"] < 65504
    p0, v0 = eng.tower_forward(boards)
    with pytest.raises(capi.AzError, match="fp16's range"):
        eng.load_weights(over)
    assert eng.weights_info()["overflow_weights"] > 0
    with pytest.raises(capi.AzError, match="fp16's range"):
        eng.load_weights(under)
    info = eng.weights_info()
    assert info["underflow_channels"] == 1 and info["overflow_weights"] == 0
    p1, v1 = eng.tower_forward(boards)                   # the weights loaded before are still in place
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    # a bf16 context takes both (fp32's exponent range)
    import torch
    torch.set_num_threads(8)
    eb = capi.Engine(N, 5, B, Cs=Cs, precision=capi.AZ_PREC_BF16)
    eb.load_weights(under)
    pb, vb = eb.tower_forward(boards)
    # ... and so does an fp16 context with the fallback switched on: same bits as the bf16 context, and it says so
    ef = capi.Engine(N, 5, B, Cs=Cs, range_fallback=1)
    ef.load_weights(under)
    info = ef.weights_info()
    assert info["fell_back"] == 1 and info["precision_in_use"] == capi.AZ_PREC_BF16
    pf, vf = ef.tower_forward(boards)
    assert np.array_equal(pf, pb) and np.array_equal(vf, vb)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, under, boards)
    assert np.abs(pf - rp).max() < 5e-3 and np.abs(vf - rv).max() < 5e-3      # bf16's own accuracy, not the 1e-3 bar
    ef.load_weights(good)                                # the next good blob is back on fp16
    assert ef.weights_info()["precision_in_use"] == capi.AZ_PREC_FP16 and ef.weights_info()["fell_back"] == 0
    pg, vg = ef.tower_forward(boards)
    assert np.array_equal(pg, p0) and np.array_equal(vg, v0)


@pytest.mark.parametrize("fused", [True, False])
def test_saturated_residual_stream_is_counted(monkeypatch, fused):
    """Weights inside fp16's range can still drive the ACTIVATIONS past 65504: the convolutions clamp there, the
    heads see it in the residual stream and az_stats.saturated counts it."""
    if not fused:
        monkeypatch.setenv("AZ_NO_FUSED", "1")
    N, B, Cs = 9, 6, synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 78)
    lay, off = {}, 0
    for name, shape in synth.blob_layout(N, 3, Cs):
        lay[name] = (off, int(np.prod(shape)))
        off += int(np.prod(shape))
    eng = capi.Engine(N, 5, B, Cs=Cs)
    eng.load_weights(blob)
    eng.tower_forward(synth.random_boards(B, N, seed=78))
    assert eng.stats()["saturated"] == 0
    for i in range(3):                                   # every block's second BatchNorm amplifies by 300
        o, n = lay["RBlock_%d.bn2.weight" % i]
        blob[o:o + n] = 300.0
    eng.load_weights(blob)
    eng.tower_forward(synth.random_boards(B, N, seed=78))
    assert eng.stats()["saturated"] > 0


@pytest.mark.parametrize("prec", [capi.AZ_PREC_BF16, capi.AZ_PREC_FP16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("N,B", [(9, 37), (15, 64), (15, 3), (19, 5), (3, 40)])
def test_tower_matches_torch_fp32(N, B, prec):
    """Ragged batch sizes (tiles that straddle boards, partial last tile)."""
    import torch
    torch.set_num_threads(8)
    Cs = synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 11 + N)
    boards = synth.random_boards(B, N, seed=N)
    eng = capi.Engine(N, min(N, 5), max(B, 4), Cs=Cs, precision=prec)
    eng.load_weights(blob)
    pol, val = eng.tower_forward(boards)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards)
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL
    assert np.allclose(pol.sum(1), 1.0, atol=1e-4)
    # a second, smaller call must not see stale rows of the first
    pol2, val2 = eng.tower_forward(boards[:2])
    assert np.abs(pol2 - rp[:2]).max() < TOL and np.abs(val2 - rv[:2]).max() < TOL


@pytest.mark.parametrize("rows", ["128", "256"])
@pytest.mark.parametrize("N,B", [(9, 300), (3, 70), (10, 5), (5, 33)])
def test_fused_tile_heights_match_torch(monkeypatch, rows, N, B):
    """Boards of at most 128 raster positions run in 128-row tiles when that saves tile rounds;
    both heights forced here, on batches with ragged last tiles and several boards per tile."""
    import torch
    torch.set_num_threads(8)
    monkeypatch.setenv("AZ_FUSED_ROWS", rows)
    Cs = synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 51 + N)
    boards = synth.random_boards(B, N, seed=50 + N)
    eng = capi.Engine(N, min(N, 5), B, Cs=Cs)
    assert eng.tower_is_fused()
    eng.load_weights(blob)
    pol, val = eng.tower_forward(boards)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards)
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL
    pol2, val2 = eng.tower_forward(boards[:3])
    assert np.array_equal(pol2, pol[:3]) and np.array_equal(val2, val[:3])


def test_per_layer_path_matches_fused_path(monkeypatch):
    """The per-convolution kernel (used for N > 15, and for C > 128 when the width is not a
    multiple of 256) against the fused kernel on a shape both can run."""
    N, B, Cs = 15, 9, synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 21)
    boards = synth.random_boards(B, N, seed=21)
    e1 = capi.Engine(N, 5, B, Cs=Cs)
    assert e1.tower_is_fused()
    e1.load_weights(blob)
    p1, v1 = e1.tower_forward(boards)
    monkeypatch.setenv("AZ_NO_FUSED", "1")
    e2 = capi.Engine(N, 5, B, Cs=Cs)
    assert not e2.tower_is_fused()
    e2.load_weights(blob)
    p2, v2 = e2.tower_forward(boards)
    assert np.abs(p1 - p2).max() < TOL and np.abs(v1 - v2).max() < TOL


@pytest.mark.parametrize("kernel", ["t256", "t256_32", "t128"])
@pytest.mark.parametrize("N,B", [(9, 6), (15, 5), (19, 3)])
def test_wide_tower_256_channels_matches_torch(monkeypatch, kernel, N, B):
    """BASELINE config 5 shape class: 256-channel blocks (per-layer path), on the 256-row kernel
    (the default: 16x16x32 MFMAs, activations as the A operand), on its 32x32x16 form with the weights
    as A, and on the 128-row kernel; batches whose last tile is partial and whose tiles straddle boards."""
    import torch
    torch.set_num_threads(8)
    if kernel == "t128":
        monkeypatch.setenv("AZ_CONV_T128", "1")
    if kernel == "t256_32":
        monkeypatch.setenv("AZ_T256_32", "1")
    Cs = [256, 256, 256]
    blob = synth.make_blob(N, 3, Cs, 31)
    boards = synth.random_boards(B, N, seed=31)
    eng = capi.Engine(N, 5, B, Cs=Cs)
    assert not eng.tower_is_fused()
    eng.load_weights(blob)
    pol, val = eng.tower_forward(boards)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards)
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL


def test_wide_tower_many_tiles_matches_128_row_kernel(monkeypatch):
    """A 256-channel tower on a batch of several hundred tiles with a ragged end (four input slices per
    convolution: three slab reloads per tile).  Checked against the 128-row kernel on the same weights
    (same operand rounding, different summation order) and, for a sample of boards, against torch fp32."""
    import torch
    torch.set_num_threads(8)
    N, B, Cs = 9, 1400, [256, 256, 256]                 # 1400 boards x 100 raster rows = 547 tiles of 256 rows
    blob = synth.make_blob(N, 3, Cs, 41)
    boards = synth.random_boards(B, N, seed=41)
    eng = capi.Engine(N, 5, B, Cs=Cs)
    eng.load_weights(blob)
    pol, val = eng.tower_forward(boards)
    eng.close()
    monkeypatch.setenv("AZ_CONV_T128", "1")
    eng2 = capi.Engine(N, 5, B, Cs=Cs)
    eng2.load_weights(blob)
    pol2, val2 = eng2.tower_forward(boards)
    assert np.abs(pol - pol2).max() < 1e-4 and np.abs(val - val2).max() < 1e-3
    pick = np.r_[0:8, 636:644, B - 8:B]                 # first tiles, the middle, the ragged end
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards[pick])
    assert np.abs(pol[pick] - rp).max() < TOL and np.abs(val[pick] - rv).max() < TOL


def test_large_board_256_row_kernel_matches_128_row_kernel(monkeypatch):
    """19x19 boards (400 raster rows: tiles straddle boards) on the 256-row x 128-channel per-layer kernel
    against the 128-row kernel on the same weights, and a sample against torch fp32."""
    import torch
    torch.set_num_threads(8)
    N, B, Cs = 19, 70, synth.NETCONFIG2                 # 70 boards x 400 rows = 110 tiles of 256 rows (ragged end)
    blob = synth.make_blob(N, 3, Cs, 61)
    boards = synth.random_boards(B, N, seed=61)
    eng = capi.Engine(N, 5, B, Cs=Cs)
    assert not eng.tower_is_fused()
    eng.load_weights(blob)
    pol, val = eng.tower_forward(boards)
    eng.close()
    monkeypatch.setenv("AZ_CONV_T128", "1")
    eng2 = capi.Engine(N, 5, B, Cs=Cs)
    eng2.load_weights(blob)
    pol2, val2 = eng2.tower_forward(boards)
    assert np.abs(pol - pol2).max() < 1e-4 and np.abs(val - val2).max() < 1e-3
    pick = np.r_[0:3, 33:36, B - 3:B]
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards[pick])
    assert np.abs(pol[pick] - rp).max() < TOL and np.abs(val[pick] - rv).max() < TOL


def test_tower_requires_weights_and_uniform_widths():
    eng = capi.Engine(9, 5, 4, Cs=synth.NETCONFIG2)
    with pytest.raises(capi.AzError):
        eng.tower_forward(synth.random_boards(2, 9))
    with pytest.raises(capi.AzError):
        capi.Engine(9, 5, 4, Cs=[128, 64, 32])


@pytest.mark.parametrize("fused", [True, False])
def test_weights_are_folded_on_the_device_and_bad_blobs_are_refused(monkeypatch, fused):
    """az_load_weights_dev (the blob already on the GPU, as after an RCCL broadcast) gives the bits of
    az_load_weights; a blob with a non-positive BatchNorm variance or a non-finite value is refused with
    AZ_E_ARG and leaves the weights in use untouched."""
    import torch
    if not fused:
        monkeypatch.setenv("AZ_NO_FUSED", "1")
    N, B, Cs = 9, 6, synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 77, phead_gain=5.0)
    boards = synth.random_boards(B, N, seed=77)
    e1 = capi.Engine(N, 5, B, Cs=Cs)
    assert e1.tower_is_fused() == fused
    e1.load_weights(blob)
    p1, v1 = e1.tower_forward(boards)
    e2 = capi.Engine(N, 5, B, Cs=Cs)
    t = torch.from_numpy(blob).cuda()
    torch.cuda.synchronize()
    e2.load_weights_dev(t.data_ptr(), blob.size)
    p2, v2 = e2.tower_forward(boards)
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
    layout = synth.blob_layout(N, 3, Cs)
    off = 0
    for name, shape in layout:
        if name == "RBlock_3.bn2.running_var":
            break
        off += int(np.prod(shape))
    bad = blob.copy()
    bad[off + 5] = -0.3                                  # sqrt of a negative variance
    with pytest.raises(capi.AzError, match="running_var"):
        e2.load_weights(bad)
    bad = blob.copy()
    bad[1234] = np.nan
    with pytest.raises(capi.AzError, match="non-finite"):
        e2.load_weights(bad)
    p3, v3 = e2.tower_forward(boards)                    # the weights loaded before are still in place
    assert np.array_equal(p3, p1) and np.array_equal(v3, v1)


@pytest.mark.parametrize("prec", [capi.AZ_PREC_FP16, capi.AZ_PREC_BF16], ids=["fp16", "bf16"])
def test_blokus_shaped_tower_matches_torch_fp32(prec):
    """20x20 board, 5 input planes, 91 policy channels (softmax over 36 400 entries), a four-entry value head:
    the per-layer tower on the shape the Blokus search feeds, against the torch fp32 statement of
    network.hpp:117-123 with `in` = 5, `out` = 91 and fc2 64 -> 4."""
    import torch
    torch.set_num_threads(8)
    N, B, Cs = 20, 7, synth.NETCONFIG2
    blob = synth.make_blob(N, 5, Cs, 123, phead_gain=3.0, fc2_gain=3.0, c_out=91, value_dim=4)
    planes = synth.random_planes(B, 5, N, seed=123)
    eng = capi.Engine(N, 5, B, Cs=Cs, precision=prec, c_in=5, c_out=91, value_dim=4, max_budget=1)
    assert not eng.tower_is_fused() and eng.weights_size() == blob.size
    eng.load_weights(blob)
    pol, val = eng.tower_forward(planes)
    assert pol.shape == (B, 91 * 400) and val.shape == (B, 4)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, planes, c_out=91, value_dim=4)
    print("blokus-shaped tower prec", prec, "policy err %.3g (max p %.3g) value err %.3g" % (np.abs(pol - rp).max(), rp.max(), np.abs(val - rv).max()))
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL
    assert np.allclose(pol.sum(1), 1.0, atol=1e-4) and np.allclose(val.sum(1), 2.0 - 4.0, atol=1e-5)
    with pytest.raises(capi.AzError, match="m,n,k search"):
        eng.search(1)                                   # this context is a tower, not an m,n,k searcher
    with pytest.raises(capi.AzError, match="holds no games"):
        eng.env_reset()
    assert eng.stats()["sims"] == 0


@pytest.mark.parametrize("c_in,c_out,vd,C", [(4, 2, 3, 192), (5, 100, 2, 128)], ids=["4in-2out-3val-192ch", "5in-100out-128ch"])
def test_odd_plane_and_channel_counts_match_torch_fp32(c_in, c_out, vd, C):
    """Plane and head counts neither game here uses: the generic stem kernel (4 planes), row blocks of the head
    GEMM that are mostly padding (2 policy channels, 192 channels in a 256-wide image) and a head with more than
    three row blocks (100 policy channels + the value channel = four blocks of 32, taken in two groups), against
    the same torch fp32 statement of network.hpp:117-123."""
    import torch
    torch.set_num_threads(8)
    N, B, Cs = 9, 5, [C, C, C]
    blob = synth.make_blob(N, c_in, Cs, 77, phead_gain=5.0, fc2_gain=3.0, c_out=c_out, value_dim=vd)
    planes = synth.random_planes(B, c_in, N, seed=77)
    eng = capi.Engine(N, 5, B, Cs=Cs, c_in=c_in, c_out=c_out, value_dim=vd, max_budget=1)
    assert not eng.tower_is_fused() and eng.weights_size() == blob.size
    eng.load_weights(blob)
    pol, val = eng.tower_forward(planes)
    assert pol.shape == (B, c_out * 81) and val.shape == (B, vd)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, planes, c_out=c_out, value_dim=vd)
    print("odd-shaped tower: policy err %.3g (max p %.3g) value err %.3g" % (np.abs(pol - rp).max(), rp.max(), np.abs(val - rv).max()))
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL
    assert np.allclose(pol.sum(1), 1.0, atol=1e-4)


def test_blokus_shaped_tower_at_batch_2048_matches_the_reference_fixture():
    """The Blokus evaluator at BASELINE configs[3]'s batch: 2048 boards of 20x20 with 5 planes, 91 policy channels.
    The first boards are those of the reference fixture (the unmodified network.hpp with in = 5, out = 91): 1e-3 at
    the full batch; the rest through properties: every policy a distribution over the 36 400 entries, and a
    permutation of the batch permutes the outputs bit for bit."""
    z = np.load(os.path.join(GOLDEN, "tower_n20_in5_out91.npz"))
    N, Cs, nb, B = int(z["N"]), [int(c) for c in z["Cs"]], int(z["B"]), 2048
    blob = synth.fixture_blob(z)
    planes = synth.random_planes(B, 5, N, seed=2048)
    planes[:nb] = z["boards"]
    eng = capi.Engine(N, 5, B, Cs=Cs, c_in=5, c_out=91, max_budget=1)
    eng.load_weights(blob)
    pol, val = eng.tower_forward(planes)
    assert np.abs(pol[:nb] - z["policy"]).max() < TOL and np.abs(val[:nb] - z["value"]).max() < TOL
    assert np.isfinite(pol).all() and (pol >= 0).all() and np.abs(pol.sum(1) - 1.0).max() < 1e-4
    assert np.abs(val.sum(1)).max() < 1e-6
    perm = np.random.default_rng(6).permutation(B)[:32]
    polp, valp = eng.tower_forward(planes[perm])
    assert np.array_equal(polp, pol[perm]) and np.array_equal(valp, val[perm])
    assert eng.stats()["saturated"] == 0


def test_masked_policy_output_has_the_bits_of_the_dense_one():
    """az_tower_forward_masked_dev on the Blokus tower's shape: the entries whose bit is set get exactly the values
    az_tower_forward_dev writes, every other entry of the output buffer is left as it was."""
    import torch
    N, B, Cs = 20, 5, synth.NETCONFIG2
    blob = synth.make_blob(N, 5, Cs, 124, phead_gain=3.0, fc2_gain=3.0, c_out=91, value_dim=4)
    planes = torch.from_numpy(synth.random_planes(B, 5, N, seed=124)).cuda()
    eng = capi.Engine(N, 5, B, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1)
    eng.load_weights(blob)
    dense = torch.empty((B, 91 * 400), dtype=torch.float32, device="cuda")
    val = torch.empty((B, 4), dtype=torch.float32, device="cuda")
    sparse = torch.full((B, 91 * 400), -5.0, dtype=torch.float32, device="cuda")
    val2 = torch.empty((B, 4), dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(3)
    want = (rng.random((B, 91, 20, 20)) < 0.01)
    want[0] = False                                        # a board that wants nothing
    want[1, 90, 19, 19] = want[1, 0, 0, 0] = True
    words = (want.astype(np.uint32) << np.arange(20, dtype=np.uint32)).sum(-1).astype(np.uint32)      # [B, 91, 20]
    wt = torch.from_numpy(words.view(np.int32)).cuda()
    torch.cuda.synchronize()
    eng.tower_forward_dev(planes.data_ptr(), B, dense.data_ptr(), val.data_ptr())
    eng.tower_forward_masked_dev(planes.data_ptr(), B, sparse.data_ptr(), val2.data_ptr(), wt.data_ptr())
    eng.sync()
    d, s_, m = dense.cpu().numpy(), sparse.cpu().numpy(), want.reshape(B, -1)
    assert m.sum() > 1000 and np.array_equal(s_[m], d[m]) and (s_[~m] == -5.0).all()
    assert np.array_equal(val.cpu().numpy(), val2.cpu().numpy())


_FORMS = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from alphazero_amd import capi, synth
z = np.load(%r)
N, Cs, B = int(z["N"]), [int(c) for c in z["Cs"]], int(z["B"])
c_in, c_out = (int(z["c_in"]), int(z["c_out"])) if "c_in" in z.files else (3, 1)
eng = capi.Engine(N, min(N, 5), B, Cs=Cs, precision=capi.AZ_PREC_FP16X3, c_in=c_in, c_out=c_out, max_budget=1)
eng.load_weights(synth.fixture_blob(z))
pol, val = eng.tower_forward(z["boards"])
np.savez(%r, pol=pol, val=val)
"""


@pytest.mark.parametrize("fixture", ["tower_n20_in5_out91.npz", "tower_n15_cfg2_spread.npz"])
def test_fp16_pair_stem_and_heads_agree_with_the_fp32_mfma_forms(fixture, tmp_path, monkeypatch):
    """The per-layer stem and heads on the 16-bit MFMAs with fp16-pair operands against the same kernels on the fp32
    MFMAs (AZ_STEM_FP32 / AZ_HEADS_FP32, read once per process: one child process each), in the compensated mode,
    where nothing behind the stem amplifies a last-bit difference: the outputs agree to 2e-6 and both hold the
    reference fixture to 1e-4."""
    import subprocess
    import sys
    monkeypatch.setenv("AZ_NO_FUSED", "1")
    root = os.path.dirname(GOLDEN[:-len("/golden")])
    path = os.path.join(GOLDEN, fixture)
    outs = []
    for tag, env in (("pair", {}), ("fp32", {"AZ_STEM_FP32": "1", "AZ_HEADS_FP32": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _FORMS % (root, path, out)], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(out))
    z = np.load(path)
    dp, dv = np.abs(outs[0]["pol"] - outs[1]["pol"]).max(), np.abs(outs[0]["val"] - outs[1]["val"]).max()
    print(fixture, "pair vs fp32 forms: policy %.3g value %.3g" % (dp, dv))
    assert dp < 2e-6 and dv < 2e-6
    for o in outs:
        assert np.abs(o["pol"] - z["policy"]).max() < 1e-4 and np.abs(o["val"] - z["value"]).max() < 1e-4


@pytest.mark.parametrize("fixture", ["tower_n20_in5_out91.npz", "tower_n15_cfg2_spread_sharp.npz", "tower_n15_c256x21_spread.npz"])
def test_compensated_mode_split_in_the_epilogue_equals_the_separate_split(fixture, tmp_path, monkeypatch):
    """AZ_PREC_FP16X3: the convolution that writes its output as the next one's [hi | lo | hi] rows against the first
    form (fp32 layer outputs, split by a kernel of its own; AZ_X3_SEPARATE_SPLIT, read once per process: one child
    process each).  The same arithmetic: the same bits, 128 and 256 channels, 15x15 and 20x20."""
    import subprocess
    import sys
    monkeypatch.setenv("AZ_NO_FUSED", "1")
    root = os.path.dirname(GOLDEN[:-len("/golden")])
    path = os.path.join(GOLDEN, fixture)
    outs = []
    for tag, env in (("epilogue", {}), ("separate", {"AZ_X3_SEPARATE_SPLIT": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", _FORMS % (root, path, out)], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["pol"], outs[1]["pol"]) and np.array_equal(outs[0]["val"], outs[1]["val"])
    z = np.load(path)
    assert np.abs(outs[0]["pol"] - z["policy"]).max() < 1e-4 and np.abs(outs[0]["val"] - z["value"]).max() < 1e-4


def test_stem_blocks_of_four_boards_equal_blocks_of_one(monkeypatch):
    """The per-layer stem takes four boards per block from 2 048 boards on (ragged last block here) and one below:
    the same boards give the same bits either way, three planes."""
    monkeypatch.setenv("AZ_NO_FUSED", "1")
    N, Cs, B = 9, synth.NETCONFIG2, 2050
    eng = capi.Engine(N, 5, B, Cs=Cs, max_budget=1)
    assert not eng.tower_is_fused()
    eng.load_weights(synth.make_blob(N, 3, Cs, 12))
    boards = synth.random_boards(B, N, seed=12)
    pol, val = eng.tower_forward(boards)
    for lo, hi in ((0, 1025), (1025, 2050)):
        p, v = eng.tower_forward(boards[lo:hi])
        assert np.array_equal(p, pol[lo:hi]) and np.array_equal(v, val[lo:hi])
    assert eng.stats()["saturated"] == 0
