"""Edge cases of the search / env entry points on the GPU, each against the oracle's
restatement of the reference's behaviour (bit-exact) or against the reference's error rule."""
import os

import numpy as np
import pytest

import azo
from alphazero_amd import capi, synth

pytestmark = pytest.mark.gpu


def _fresh(N, K, G, budget, **kw):
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=max(budget, 8), **kw)
    noise = np.stack([azo.dirichlet(40 + g, N * N)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.env_reset()
    return eng, noise


@pytest.mark.parametrize("budget", [-3, 0, 1, 2])
def test_budget_below_one_still_runs_one_simulation(budget):
    """mcts.hpp:161-174: the do/while body runs once even for iter_budget <= 0."""
    N, K, G = 5, 4, 3
    eng, noise = _fresh(N, K, G, budget)
    out = eng.search(budget)
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        ref = m.search([], budget, noise[g])
        assert np.array_equal(out[g], ref)
        assert eng.tree_canonical(g) == azo.canonical_tree(m.dump())
        assert m.root_n() == 1 + max(budget, 1)


def test_repeated_search_on_the_same_root_accumulates():
    """generate.cpp:77,84: warm-up search and the move's search share the root (find() hit with
    an identical history): visit counts add up."""
    N, K, G, budget = 7, 5, 2, 60
    eng, noise = _fresh(N, K, G, budget)
    orc = [azo.MCTS(N, K, 3.0, 3.0, "hash") for _ in range(G)]
    for rep in range(3):
        out = eng.search(budget)
        for g in range(G):
            ref = orc[g].search([], budget, noise[g])
            assert np.array_equal(out[g], ref)
            assert orc[g].root_n() == 1 + (rep + 1) * budget
            assert eng.tree_canonical(g) == azo.canonical_tree(orc[g].dump())


def test_tree_clear_matches_oracle_clear():
    """MCTS::clear (mcts.hpp:462-465) drops the tree: the next search starts from a fresh root
    and draws fresh root noise."""
    N, K, G, budget = 5, 4, 2, 40
    eng, noise = _fresh(N, K, G, budget)
    first = eng.search(budget)
    eng.tree_clear()
    again = eng.search(budget)
    assert np.array_equal(first, again)
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        m.search([], budget, noise[g])
        m.clear()
        assert np.array_equal(again[g], m.search([], budget, noise[g]))


def test_search_on_a_finished_game_returns_zeros_and_leaves_others_alone():
    N, K, G, budget = 3, 3, 2, 30
    eng, noise = _fresh(N, K, G, budget)
    # game 0: player 0 wins on the top row; game 1: two moves only
    eng.env_load([[0, 3, 1, 4, 2], [4, 0]])
    out = eng.search(budget)
    assert not out[0].any()
    m = azo.MCTS(N, K, 3.0, 3.0, "hash")
    assert np.array_equal(out[1], m.search([4, 0], budget, noise[1]))
    with pytest.raises(ValueError):
        azo.MCTS(N, K, 3.0, 3.0, "hash").search([0, 3, 1, 4, 2], budget, noise[0])   # the oracle refuses it too


def test_env_load_rejects_illegal_histories():
    N, K, G = 3, 3, 1
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=8)
    with pytest.raises(capi.AzError, match="illegal move"):
        eng.env_load([[4, 4]])                          # occupied cell
    with pytest.raises(capi.AzError, match="illegal move"):
        eng.env_load([[9]])                             # off the board
    with pytest.raises(capi.AzError, match="after the game ended"):
        eng.env_load([[0, 3, 1, 4, 2, 5]])              # move after a win


def test_arena_overflow_restarts_the_search_from_a_fresh_root():
    """A tree that outgrows its arena is dropped and its search starts over from a fresh root (what
    __make_root does on a find() miss, mcts.hpp:392-414): the slot keeps going, the event is counted."""
    N, K, G, budget = 7, 5, 3, 64
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, node_cap=budget + 6)
    noise = np.stack([azo.dirichlet(70 + g, N * N)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.env_reset()
    first = eng.search(budget)                       # 65 nodes: fits
    assert eng.stats()["tree_restarts"] == 0
    again = eng.search(budget)                       # same root keeps accumulating (Q9) -> the arena fills -> restart
    st = eng.stats()
    assert st["tree_restarts"] == G and st["error_games"] == 0
    assert np.array_equal(again, first)              # the restarted search is a fresh 64-simulation search
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        assert np.array_equal(again[g], m.search([], budget, noise[g]))
        assert eng.tree_canonical(g) == azo.canonical_tree(m.dump())
    # an arena that cannot hold even one search is refused up front instead of restarting for ever
    with pytest.raises(capi.AzError, match="tree nodes per game"):
        capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=64, node_cap=16).search(64)
    with pytest.raises(capi.AzError, match="tree nodes per game"):
        capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=8).search(4000)


def test_search_is_deterministic_across_contexts_and_slots():
    """Same position, same noise: every slot of every context returns the same bits."""
    N, K, G, budget = 9, 5, 4, 120
    noise = np.tile(azo.dirichlet(5, N * N)[0], (G, 1))
    outs = []
    for _ in range(2):
        eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget)
        eng.set_noise(noise)
        eng.env_load([[40, 41]] * G)
        outs.append(eng.search(budget))
        eng.close()
    assert np.array_equal(outs[0], outs[1])
    assert all(np.array_equal(outs[0][0], outs[0][g]) for g in range(G))


def test_replay_drain_in_small_pieces_keeps_every_row():
    """Draining with a buffer smaller than the ring's content hands over the same rows as one big
    drain (the rows that stay are moved to the front on the device)."""
    N, K, G, budget = 3, 3, 8, 12

    def play():
        eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, seed=11)
        eng.selfplay_begin(budget)
        eng.selfplay_run(300)
        return eng

    whole = play().replay_drain()
    assert len(whole[0]) > 40
    eng = play()
    parts = []
    for cap in (3, 1, 7, 2, 1 << 16):
        parts.append(eng.replay_drain(cap))
    assert len(parts[0][0]) == 3 and len(parts[2][0]) == 7
    # games that finish in the same round reserve their ring rows in launch order, which differs from
    # run to run: compare the rows as a multiset (a bad tail move would duplicate or lose rows)
    def rows(t):
        return sorted(t[0][i].tobytes() + t[1][i:i + 1].tobytes() + t[2][i].tobytes() + t[3][i].tobytes()
                      for i in range(len(t[0])))
    joined = tuple(np.concatenate([p[i] for p in parts]) for i in range(4))
    assert rows(joined) == rows(whole)
    assert len(eng.replay_drain()[0]) == 0


def test_null_arguments_are_refused_not_dereferenced():
    lib = capi.load()
    assert lib.az_sync(None) == capi.AZ_E_ARG
    assert lib.az_get_stats(None, None) == capi.AZ_E_ARG
    assert lib.az_selfplay_batch(None) == 0
    eng = capi.Engine(3, 3, 1, eval_mode=capi.AZ_EVAL_HASH, max_budget=8)
    assert lib.az_env_step(eng.h, None, None, None) == capi.AZ_E_ARG
    assert b"null argument" in lib.az_last_error()


@pytest.mark.parametrize("L", [1, 3])
def test_traffic_tallies_follow_from_the_tree(L):
    """az_stats.descents / tree_levels / children_scanned / children_expanded are what the tree
    kernel's algorithmic bytes are computed from.  After one search from a fresh root every one of
    them follows from the finished tree: each descent adds 1 to n of every edge on its path, a
    level scans the whole child block of the node it passes, an expansion writes one block."""
    N, K, G, budget = 7, 5, 3, 150
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, leaves_per_round=L)
    eng.env_load([[24, 25], [], [0, 48, 6]])
    eng.search(budget)
    st = eng.stats()
    levels = scanned = expanded = 0
    for g in range(G):
        d = eng.tree_dump(g)
        expanded += len(d["n"])
        levels += int(d["n"].sum())
        for fe, k in zip(d["node_first"], d["node_count"]):
            scanned += int(k) * int(d["n"][fe:fe + k].sum())
    assert st["descents"] == G * budget
    assert st["tree_levels"] == levels
    assert st["children_scanned"] == scanned
    assert st["children_expanded"] == expanded


def test_largest_board_and_longest_line():
    """N = 21 is the largest board the context accepts (N*N <= 448 path entries), K = 8 the longest
    winning line: tree parity against the oracle, tower parity against torch on the per-layer path."""
    import torch
    from test_gpu_tower import TOL, torch_reference
    N, K, G, budget = 21, 8, 2, 90
    eng, noise = _fresh(N, K, G, budget)
    eng.env_load([[220, 221, 199], []])
    out = eng.search(budget)
    for g, hist in enumerate([[220, 221, 199], []]):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        assert np.array_equal(out[g], m.search(hist, budget, noise[g]))
        assert eng.tree_canonical(g) == azo.canonical_tree(m.dump())
    with pytest.raises(capi.AzError, match="unsupported"):
        capi.Engine(22, 5, 1, eval_mode=capi.AZ_EVAL_HASH, max_budget=8)
    with pytest.raises(capi.AzError, match="win length"):
        capi.Engine(9, 9, 1, eval_mode=capi.AZ_EVAL_HASH, max_budget=8)
    torch.set_num_threads(8)
    Cs = synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 5)
    boards = synth.random_boards(3, N, seed=5)
    te = capi.Engine(N, K, 4, Cs=Cs)
    assert not te.tower_is_fused()
    te.load_weights(blob)
    pol, val = te.tower_forward(boards)
    with torch.no_grad():
        rp, rv = torch_reference(N, Cs, blob, boards)
    assert np.abs(pol - rp).max() < TOL and np.abs(val - rv).max() < TOL


def test_device_dirichlet_noise_follows_the_reference_sampler():
    """The device's dirichlet(n, alpha = 0.05) (util.h:42-58: normalised std::gamma_distribution<float> draws) is
    only defined as a distribution (the reference seeds from random_device).  alpha = 0.05 is the hard corner of
    a gamma sampler (almost all mass near zero, a few components carry the sum), so the device draws of 4096
    fresh roots are compared with the oracle's std::gamma_distribution draws: exact simplex constraint, the
    moments of Dirichlet(alpha 1_k), and a two-sample Kolmogorov-Smirnov distance on the components and on the
    largest component per draw."""
    from scipy import stats
    N, K, G = 9, 5, 4096
    A = N * N
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=4, noise_mode=capi.AZ_NOISE_DEVICE, seed=123)
    eng.env_reset()
    eng.search(1)                                       # fresh roots: every game draws its noise
    dev = eng.get_noise().astype(np.float64)
    st = 99
    ref = np.empty((G, A))
    for g in range(G):
        x, st = azo.dirichlet(st, A)
        ref[g] = x
    assert np.isfinite(dev).all() and (dev >= 0).all()
    assert np.abs(dev.sum(1) - 1.0).max() < 1e-5
    assert len({tuple(np.round(r, 6)) for r in dev[:64]}) == 64          # games draw different noise
    alpha, k = 0.05, A
    var = (k - 1) / (k * k * (k * alpha + 1.0))                          # component of Dirichlet(alpha 1_k)
    for name, x in (("device", dev), ("oracle", ref)):
        assert abs(x.mean() - 1.0 / k) < 1e-9, name
        assert abs(x.var() / var - 1.0) < 0.05, (name, x.var(), var)
    # each position k of the draw is the same distribution (no lane- or index-dependent bias)
    col = dev.mean(0)
    assert np.abs(col - 1.0 / k).max() < 6 * np.sqrt(var / G)
    d_comp = stats.ks_2samp(dev.ravel(), ref.ravel()).statistic
    d_max = stats.ks_2samp(dev.max(1), ref.max(1)).statistic
    frac_tiny = ((dev < 1e-12).mean(), (ref < 1e-12).mean())
    print("KS components %.4f, KS max-component %.4f, fraction below 1e-12: device %.4f oracle %.4f" % ((d_comp, d_max) + frac_tiny))
    assert d_comp < 0.01
    assert d_max < 0.05
    assert abs(frac_tiny[0] - frac_tiny[1]) < 0.01


def test_overlapped_row_hand_over_gives_the_same_rows():
    """az_replay_flip / az_replay_drain_retired (rows of move k handed over while move k + 1 is already queued)
    against the plain az_replay_drain on the same seeds: the two engines play the same games, so the rows --
    compared as sorted sets, finished games of one round land in the ring in no fixed order -- are the same, the
    retired drain never returns a row of a later flip period, and nothing is lost or duplicated."""
    N, K, G, budget, moves = 5, 4, 24, 16, 40
    Cs = [16, 16, 16]
    blob = synth.make_blob(N, 3, Cs, 3, phead_gain=4.0)

    def rows_key(cells, tm, pol, rew):
        return sorted((cells[i].tobytes(), int(tm[i]), pol[i].tobytes(), rew[i].tobytes()) for i in range(len(tm)))
    a = capi.Engine(N, K, G, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=4)
    b = capi.Engine(N, K, G, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=4)
    for e in (a, b):
        e.load_weights(blob)
        e.selfplay_begin(budget)
    plain = []
    for _ in range(moves):
        a.selfplay_run(budget)
        plain += rows_key(*a.replay_drain())
    over, per_period = [], []
    b.selfplay_run(budget)
    for _ in range(moves - 1):
        b.replay_flip()
        b.selfplay_run(budget)                             # queued behind the retired rounds
        got = rows_key(*b.replay_drain_retired())
        per_period.append(len(got))
        over += got
    over += rows_key(*b.replay_drain())                    # the rest: both rings, after the stream has drained
    assert len(plain) > 100 and sorted(plain) == sorted(over)
    assert sum(per_period) > 0 and b.stats()["replay_rows"] == 0 and b.stats()["replay_dropped"] == 0
    sa, sb = a.stats(), b.stats()
    assert (sa["games"], sa["moves"], sa["sims"]) == (sb["games"], sb["moves"], sb["sims"])


def test_bench_with_eight_ranks_prints_one_line_that_adds_up():
    """bench.py as the driver launches it for N = 8 (python -m torch.distributed.run, one rank per GPU), rehearsed on
    this box with the eight ranks sharing the one GPU and the collectives on gloo (AZ_BENCH_BACKEND): exactly one
    JSON line comes out, from rank 0; `value` = the sum of the ranks' device-side simulation counters over the
    slowest rank's time; the single-GPU extras (CPU baseline, other configs) stay out."""
    import json
    import subprocess
    import sys
    from test_wire import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (eight processes on one GPU: the persistent pair, which wants the GPU to itself, stays off whatever the environment says)
    env = dict(os.environ, AZ_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", AZ_PAIR="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--games", "32", "--budget", "32", "--board", "9", "--full-games", "0", "--no-lockstep-probe"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["metric"] == "mcts_simulations_per_sec" and r["scaling"] == "weak" and r["steps"] == 2
    assert "cpu_baseline" not in r and "other_configs" not in r
    pr = r["per_rank"]
    assert len(pr["sims"]) == 8 and all(s == 2 * 32 * 32 for s in pr["sims"])      # every rank: 32 games x 32 sims x 2 moves
    assert abs(r["value"] - sum(pr["sims"]) / max(pr["seconds"])) < 1e-6 * r["value"]
    assert abs(r["ms_per_step"] - max(pr["seconds"]) / 2 * 1e3) < 1e-6 * r["ms_per_step"]
