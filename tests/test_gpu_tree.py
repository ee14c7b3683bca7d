"""GPU parity of the env / tree kernels against the CPU oracle and against the
fixtures produced by the unmodified reference MCTS.  Everything is bit-exact."""
import glob
import os

import numpy as np
import pytest

import azo
from alphazero_amd import capi

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MCTS_FIX = sorted(glob.glob(os.path.join(GOLDEN, "mcts_*.npz")))


def test_env_random_playouts_bit_exact():
    N, K, G = 9, 5, 64
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH)
    rng = np.random.default_rng(1)
    cells = [azo.env_reset(N)[0] for _ in range(G)]
    tm = [0] * G
    over = [False] * G
    eng.env_reset()
    for step in range(N * N):
        acts = np.full(G, -1, np.int32)
        for g in range(G):
            if not over[g]:
                acts[g] = rng.choice(azo.env_legal(N, cells[g]))
        if (acts < 0).all():
            break
        rew, done = eng.env_step(acts)
        for g in range(G):
            if acts[g] >= 0:
                cells[g], tm[g], r, d = azo.env_step(N, K, cells[g], tm[g], acts[g])
                assert np.array_equal(rew[g], r) and bool(done[g]) == bool(d)
                over[g] = bool(d)
        planes = eng.env_boards()
        mask = eng.env_legal_mask()
        for g in range(G):
            assert np.array_equal(planes[g], azo.env_planes(N, cells[g], tm[g]))
            want = (cells[g] == 0) & (not over[g])
            assert np.array_equal(mask[g].reshape(-1).astype(bool), want)
    assert all(over)
    with pytest.raises(capi.AzError):
        eng.env_step(np.zeros(G, np.int32))       # finished games reject moves


def test_hash_evaluator_matches_oracle():
    N, G = 7, 8
    eng = capi.Engine(N, 5, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=4)
    eng.env_reset()
    pol = eng.search(1)
    # root n = 2 after one simulation: policy = 1/2 on the first maximal child
    assert np.allclose(pol.sum(1), 0.5)


@pytest.mark.parametrize("path", MCTS_FIX, ids=[os.path.basename(p) for p in MCTS_FIX])
def test_search_matches_reference_fixture(path):
    """External-evaluator search fed with the reference evaluator's recorded
    answers reproduces the reference tree bit for bit, move after move."""
    z = np.load(path)
    N, K, budget, steps, _ = [int(x) for x in z["meta"]]
    A = N * N
    table = {bytes(p): (pol, val) for p, pol, val in zip(z["table_planes"], z["table_policy"], z["table_value"])}
    G = 2                                     # two identical games: slots must not interact
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_EXTERNAL, max_budget=budget)
    eng.set_noise(np.tile(z["noise"], (G, 1)))
    eng.env_reset()
    for s in range(steps):
        eng.search_begin(budget)
        while True:
            planes, pending, running = eng.search_select()
            if running == 0:
                break
            pol = np.zeros((G, A), np.float32)
            val = np.zeros((G, 2), np.float32)
            for g in range(G):
                if pending[g] in (1, 2):
                    pol[g], val[g] = table[planes[g].tobytes()]
            eng.search_apply(pol, val)
        out, root_n = eng.search_result()
        ref = list(zip(*[z["step%d_%s" % (s, k)].tolist() for k in ("parent", "action", "n", "v", "p", "kind")]))
        for g in range(G):
            assert np.array_equal(out[g], z["policies"][s])
            assert root_n[g] == z["root_n"][s]
            got = eng.tree_canonical(g)
            assert len(got) == len(ref)
            assert got == [tuple(r) for r in ref]
        _, done = eng.env_step(np.full(G, int(z["moves"][s]), np.int32))


@pytest.mark.parametrize("N,K,budget,moves", [(3, 3, 64, 9), (9, 5, 200, 6), (15, 5, 300, 4)])
def test_search_matches_oracle_hash_evaluator(N, K, budget, moves):
    """Fused device search (hash pseudo-network) == oracle, including root
    reuse across moves, for games that follow different lines."""
    G = 6
    A = N * N
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget)
    noise = np.stack([azo.dirichlet(100 + g, A)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.env_reset()
    orc = [azo.MCTS(N, K, 3.0, 3.0, "hash") for _ in range(G)]
    hist = [[] for _ in range(G)]
    alive = [True] * G
    for mv in range(moves):
        out = eng.search(budget)
        acts = np.full(G, -1, np.int32)
        for g in range(G):
            if not alive[g]:
                continue
            ref = orc[g].search(hist[g], budget, noise[g])
            assert np.array_equal(out[g], ref), (mv, g)
            assert eng.tree_canonical(g) == azo.canonical_tree(orc[g].dump())
            # different games take different moves: g-th most visited child
            order = np.argsort(-ref, kind="stable")
            acts[g] = order[min(g, 2)] if ref[order[min(g, 2)]] > 0 else order[0]
        _, done = eng.env_step(acts)
        for g in range(G):
            if alive[g]:
                hist[g].append(int(acts[g]))
                alive[g] = not bool(done[g])


def test_selfplay_matches_oracle_games():
    """Device-side generate loop (hash pseudo-network): every finished game's
    rows equal the oracle's replay of the same slot."""
    N, K, budget, G = 5, 4, 48, 8
    A = N * N
    seed = 77
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, seed=seed)
    noise = np.stack([azo.dirichlet(500 + g, A)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.selfplay_begin(budget)
    eng.selfplay_run((2 * budget + 1 + (A - 1) * budget) * 2 + 8)    # >= 2 full games per slot
    cells, tm, pol, rew = eng.replay_drain()
    st = eng.stats()
    assert st["games"] >= 2 * G and st["error_games"] == 0
    # oracle: play the same number of games per slot, collect rows as a multiset
    want = []
    n_games = {g: 0 for g in range(G)}
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        avg, gc, sc = 25.0, 0, 0
        for _ in range(2):
            r = m.selfplay_game(budget, seed, g, noise[g], avg, gc, sc)
            avg, gc, sc = r["avg_len"], r["game_count"], r["sample_counter"]
            for i in range(r["n_moves"]):
                want.append((r["cells"][i].tobytes(), int(r["to_move"][i]), r["policy"][i].tobytes(), r["reward"].tobytes()))
    got = {(cells[i].tobytes(), int(tm[i]), pol[i].tobytes(), rew[i].tobytes()) for i in range(len(tm))}
    missing = [w for w in want if w not in got]
    assert not missing, "%d of %d oracle rows not produced by the device" % (len(missing), len(want))


@pytest.mark.parametrize("L", [2, 4, 8])
@pytest.mark.parametrize("N,K,budget,moves", [(3, 3, 64, 9), (9, 5, 203, 5)])
def test_multi_leaf_search_matches_oracle(N, K, budget, moves, L):
    """L descents per tree per round (the reference's nthreads = L in a fixed
    order): trees and policies equal the oracle's restatement of the same
    order, across root reuse, with budgets that L does not divide."""
    G = 5
    A = N * N
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, leaves_per_round=L)
    noise = np.stack([azo.dirichlet(300 + g, A)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.env_reset()
    orc = [azo.MCTS(N, K, 3.0, 3.0, "hash") for _ in range(G)]
    hist = [[] for _ in range(G)]
    alive = [True] * G
    for mv in range(moves):
        out = eng.search(budget)
        acts = np.full(G, -1, np.int32)
        for g in range(G):
            if not alive[g]:
                continue
            ref = orc[g].search_k(hist[g], budget, L, noise[g])
            assert np.array_equal(out[g], ref), (mv, g)
            assert eng.tree_canonical(g) == azo.canonical_tree(orc[g].dump())
            order = np.argsort(-ref, kind="stable")
            acts[g] = order[min(g, 2)] if ref[order[min(g, 2)]] > 0 else order[0]
        _, done = eng.env_step(acts)
        for g in range(G):
            if alive[g]:
                hist[g].append(int(acts[g]))
                alive[g] = not bool(done[g])


def test_multi_leaf_external_evaluator():
    """The select/apply protocol with L = 3 slots per game: the caller answers
    slot g*L+k; result equals the fused search."""
    N, K, G, L, budget = 7, 5, 3, 3, 100
    A = N * N
    noise = np.stack([azo.dirichlet(900 + g, A)[0] for g in range(G)])
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_EXTERNAL, max_budget=budget, leaves_per_round=L)
    eng.set_noise(noise)
    eng.env_reset()
    eng.search_begin(budget)
    batch_sizes = []
    while True:
        planes, pending, running = eng.search_select()
        if running == 0:
            break
        pol = np.zeros((G * L, A), np.float32)
        val = np.zeros((G * L, 2), np.float32)
        for s in range(G * L):
            if pending[s] in (1, 2):
                pol[s], val[s] = azo.eval_hash(N, planes[s])
        batch_sizes.append(int(np.isin(pending, (1, 2)).sum()))
        eng.search_apply(pol, val)
    out, root_n = eng.search_result()
    assert max(batch_sizes) > G                       # more than one leaf per game in a round
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        assert np.array_equal(out[g], m.search_k([], budget, L, noise[g]))
        assert root_n[g] == m.root_n()
        assert eng.tree_canonical(g) == azo.canonical_tree(m.dump())


def test_multi_leaf_selfplay_matches_oracle_games():
    N, K, budget, G, L = 5, 4, 50, 6, 4
    A = N * N
    seed = 91
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, seed=seed, leaves_per_round=L)
    noise = np.stack([azo.dirichlet(700 + g, A)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.selfplay_begin(budget)
    eng.selfplay_run((2 * budget + 1 + (A - 1) * budget) * 2 + 8)    # generous: collisions only shorten rounds' work
    cells, tm, pol, rew = eng.replay_drain()
    st = eng.stats()
    assert st["games"] >= 2 * G and st["error_games"] == 0
    want = []
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        m.set_leaves(L)
        avg, gc, sc = 25.0, 0, 0
        for _ in range(2):
            r = m.selfplay_game(budget, seed, g, noise[g], avg, gc, sc)
            avg, gc, sc = r["avg_len"], r["game_count"], r["sample_counter"]
            for i in range(r["n_moves"]):
                want.append((r["cells"][i].tobytes(), int(r["to_move"][i]), r["policy"][i].tobytes(), r["reward"].tobytes()))
    got = {(cells[i].tobytes(), int(tm[i]), pol[i].tobytes(), rew[i].tobytes()) for i in range(len(tm))}
    missing = [w for w in want if w not in got]
    assert not missing, "%d of %d oracle rows not produced by the device" % (len(missing), len(want))


@pytest.mark.parametrize("N,K,budget,L", [(3, 3, 30, 1), (4, 3, 40, 3), (6, 4, 64, 2)])
def test_selfplay_many_slots_match_oracle(N, K, budget, L):
    """A wider sweep of the device-side generate loop: 48 slots, three games each, small boards
    where wins, draws, terminal re-visits, collisions of same-round descents and dropped trees
    are all frequent.  Every row of every finished game equals the oracle's."""
    G, games = 48, 3
    A = N * N
    seed = 1234 + N
    eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, seed=seed, leaves_per_round=L)
    noise = np.stack([azo.dirichlet(9000 + g, A)[0] for g in range(G)])
    eng.set_noise(noise)
    eng.selfplay_begin(budget)
    got = set()
    need = (2 * budget + 2 + A * (budget + 1)) * games + 16      # rounds after which every slot has played `games` games
    done = 0
    while done < need:                                           # short games finish fast: drain as we go
        eng.selfplay_run(4 * budget)
        done += 4 * budget
        cells, tm, pol, rew = eng.replay_drain(1 << 16)
        got.update((cells[i].tobytes(), int(tm[i]), pol[i].tobytes(), rew[i].tobytes()) for i in range(len(tm)))
    st = eng.stats()
    assert st["games"] >= games * G and st["error_games"] == 0 and st["replay_dropped"] == 0
    missing = total = 0
    for g in range(G):
        m = azo.MCTS(N, K, 3.0, 3.0, "hash")
        m.set_leaves(L)
        avg, gc, sc = 25.0, 0, 0
        for _ in range(games):
            r = m.selfplay_game(budget, seed, g, noise[g], avg, gc, sc)
            avg, gc, sc = r["avg_len"], r["game_count"], r["sample_counter"]
            for i in range(r["n_moves"]):
                total += 1
                missing += (r["cells"][i].tobytes(), int(r["to_move"][i]), r["policy"][i].tobytes(), r["reward"].tobytes()) not in got
    assert missing == 0, "%d of %d oracle rows not produced by the device" % (missing, total)
