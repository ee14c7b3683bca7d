"""The persistent tower / tree pair (az_selfplay_run_pair) against the lockstep rounds (az_selfplay_run): the same
games bit for bit -- counters, whole tree arenas, replay rows -- and a hand-over wait that runs out drains both
kernels and is reported, instead of hanging the GPU."""
import os

import numpy as np
import pytest

from alphazero_amd import capi, synth

pytestmark = pytest.mark.gpu


def _rows_key(cells, tm, pol, rew):
    return sorted((cells[i].tobytes(), int(tm[i]), pol[i].tobytes(), rew[i].tobytes()) for i in range(len(tm)))


def _engine(N, K, G, budget, seed=5, chain=0):
    e = capi.Engine(N, K, G, Cs=synth.NETCONFIG2, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=seed)
    e.load_weights(synth.make_blob(N, 3, synth.NETCONFIG2, 0))
    e.selfplay_chain(chain)
    e.selfplay_begin(budget)
    return e


# (N, K, games, sims/move, lockstep rounds, the same rounds as pair launches)
CASES = [
    (9, 5, 24, 32, 40, [1, 7, 32]),          # 128-row tiles, launches of one round and of many
    (15, 5, 16, 24, 400, [400]),             # 256-row tiles = one game per workgroup; 240 moves with re-roots
    (15, 5, 300, 40, 200, [50, 150]),        # more games than CUs: workgroups and tree waves take two games each
    (5, 4, 37, 16, 300, [100, 200]),         # seven boards per tile, ragged last tile, whole games and restarts
    (15, 5, 1100, 16, 96, [40, 56]),         # four to five games per workgroup and tree wave
]


@pytest.mark.parametrize("chain", [0, 5])
@pytest.mark.parametrize("N,K,G,budget,rounds,chunks", CASES)
def test_pair_plays_the_same_games_as_lockstep_rounds(N, K, G, budget, rounds, chunks, chain):
    a, b = _engine(N, K, G, budget, chain=chain), _engine(N, K, G, budget, chain=chain)
    try:
        a.selfplay_run(rounds)
        a.sync()
        for ch in chunks:
            b.selfplay_run_pair(ch)
        b.sync()
        sa, sb = a.stats(), b.stats()
        assert sa == sb
        assert sa["rounds"] == rounds and sa["error_games"] == 0
        for g in range(G):
            da, db = a.tree_dump(g), b.tree_dump(g)
            for k in da:
                assert np.array_equal(da[k], db[k]), "tree of game %d differs in %s" % (g, k)
        ra, rb = _rows_key(*a.replay_drain()), _rows_key(*b.replay_drain())
        assert ra == rb
        if N == 5:
            assert sa["games"] > 0 and len(ra) > 0
        pr = b.pair_read()
        assert 0.0 < pr["tree_step_avg_us"] < 1000.0
    finally:
        a.close()
        b.close()


def test_pair_and_lockstep_rounds_mix_on_one_context():
    N, K, G, budget = 9, 5, 40, 24
    a, b = _engine(N, K, G, budget), _engine(N, K, G, budget)
    try:
        a.selfplay_run(150)
        b.selfplay_run(20)
        b.selfplay_run_pair(60)
        b.selfplay_run(30)
        b.selfplay_run_pair(40)
        a.sync()
        b.sync()
        assert a.stats() == b.stats()
        for g in range(G):
            da, db = a.tree_dump(g), b.tree_dump(g)
            for k in da:
                assert np.array_equal(da[k], db[k])
        assert _rows_key(*a.replay_drain()) == _rows_key(*b.replay_drain())
    finally:
        a.close()
        b.close()


def test_pair_needs_the_one_launch_tower_and_one_leaf_per_round():
    e = capi.Engine(9, 5, 8, Cs=synth.NETCONFIG2, max_budget=16, noise_mode=capi.AZ_NOISE_DEVICE, leaves_per_round=2)
    e.load_weights(synth.make_blob(9, 3, synth.NETCONFIG2, 0))
    e.selfplay_begin(16)
    with pytest.raises(capi.AzError, match="leaves_per_round"):
        e.selfplay_run_pair(4)
    e.close()
    e = capi.Engine(9, 5, 8, eval_mode=capi.AZ_EVAL_HASH, max_budget=16, noise_mode=capi.AZ_NOISE_DEVICE)
    e.selfplay_begin(16)
    with pytest.raises(capi.AzError, match="one launch"):
        e.selfplay_run_pair(4)
    e.close()


def test_a_wait_that_runs_out_drains_both_kernels_and_is_reported():
    """Without its tree kernel the tower's first wait can never be answered: after AZ_PAIR_TIMEOUT_MS every
    workgroup leaves, the launch ends, and the context says what happened until self-play is begun again."""
    N, K, G, budget = 9, 5, 24, 16
    e = _engine(N, K, G, budget)
    old = {k: os.environ.get(k) for k in ("AZ_PAIR_FLAGS", "AZ_PAIR_TIMEOUT_MS")}
    try:
        os.environ["AZ_PAIR_FLAGS"] = "4"                    # launch no tree kernel
        os.environ["AZ_PAIR_TIMEOUT_MS"] = "30"
        e.selfplay_run_pair(5)
        with pytest.raises(capi.AzError, match="ran out"):
            e.sync()
        with pytest.raises(capi.AzError, match="ran out"):
            e.stats()
        os.environ["AZ_PAIR_FLAGS"] = "0"
        os.environ["AZ_PAIR_TIMEOUT_MS"] = "4000"
        e.selfplay_begin(budget)                             # a fresh start clears the flag
        e.selfplay_run_pair(40)
        e.sync()
        ref = _engine(N, K, G, budget)
        ref.selfplay_run(40)
        ref.sync()
        assert e.stats()["sims_backed_up"] == ref.stats()["sims_backed_up"] > 0
        ref.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        e.close()


def test_chained_terminal_simulations_play_the_same_games_in_fewer_rounds():
    """A simulation that ends in a stored terminal leaf needs no evaluation; run back to back inside a round
    (az_selfplay_chain) such simulations change how many rounds a search takes, never what it finds: every game
    the plain rounds finish is finished, row for row, by the chained rounds in the same number of rounds."""
    N, K, G, budget, rounds = 5, 4, 37, 32, 600
    a, b = _engine(N, K, G, budget, chain=0), _engine(N, K, G, budget, chain=6)
    try:
        a.selfplay_run(rounds)
        b.selfplay_run(rounds)
        a.sync()
        b.sync()
        sa, sb = a.stats(), b.stats()
        assert sa["rounds"] == sb["rounds"] == rounds
        assert sa["sims_backed_up"] <= rounds * G + G
        assert sb["sims_backed_up"] > 1.05 * sa["sims_backed_up"]          # 5x5, four in a row: terminal leaves are common
        assert sb["evals"] <= rounds * G + G and sb["games"] >= sa["games"] > 20
        assert sa["error_games"] == sb["error_games"] == 0
        ra, rb = _rows_key(*a.replay_drain()), _rows_key(*b.replay_drain())
        assert len(ra) > 200
        have = {}
        for r in rb:
            have[r] = have.get(r, 0) + 1
        for r in ra:
            assert have.get(r, 0) > 0, "a row of the plain rounds is missing from the chained rounds"
            have[r] -= 1
    finally:
        a.close()
        b.close()


def test_chain_needs_one_leaf_per_round():
    e = capi.Engine(9, 5, 8, eval_mode=capi.AZ_EVAL_HASH, max_budget=16, leaves_per_round=2)
    with pytest.raises(capi.AzError, match="leaves_per_round"):
        e.selfplay_chain(3)
    e.selfplay_chain(0)
    e.close()


def test_pair_launches_of_two_contexts_on_one_gpu_take_turns():
    """Two contexts that both queue pair launches without waiting for each other: each pair wants the whole chip, so
    the library makes them take turns (the host waits for the other context's launch); both finish and play what the lockstep rounds play."""
    N, K, G, budget, rounds = 15, 5, 300, 24, 120
    a, b, ref = _engine(N, K, G, budget, seed=1), _engine(N, K, G, budget, seed=2), _engine(N, K, G, budget, seed=2)
    try:
        for _ in range(3):
            a.selfplay_run_pair(rounds // 3)
            b.selfplay_run_pair(rounds // 3)
        a.sync()
        b.sync()
        ref.selfplay_run(rounds)
        ref.sync()
        assert b.stats() == ref.stats()
        assert a.stats()["sims_backed_up"] == ref.stats()["sims_backed_up"]
        for g in (0, 7, G - 1):
            db, dr = b.tree_dump(g), ref.tree_dump(g)
            for k in db:
                assert np.array_equal(db[k], dr[k])
    finally:
        a.close()
        b.close()
        ref.close()


_FRESH = r"""
import sys
sys.path.insert(0, %r)
from alphazero_amd import capi, synth
N, K, G, budget = 15, 5, 300, 24
def eng(seed):
    e = capi.Engine(N, K, G, Cs=synth.NETCONFIG2, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=seed)
    e.load_weights(synth.make_blob(N, 3, synth.NETCONFIG2, 0))
    e.selfplay_begin(budget)
    return e
es = [eng(s) for s in (11, 12, 13)]
ref = eng(12)
for _ in range(3):
    for e in es:
        e.selfplay_run_pair(20)
        e.selfplay_run_pair(20)
ref.selfplay_run(120)
for e in es + [ref]:
    e.sync()
assert es[1].stats() == ref.stats()
print("fresh ok", es[0].stats()["sims_backed_up"])
"""


def test_pair_launches_of_three_contexts_queued_from_a_fresh_process():
    """The first thing a process does on the GPU: three contexts queue pair launches in turn without a host wait of
    their own.  (The hardware queues the streams get depend on what the process made before; with the queues a fresh
    process gets, a wait on the GPU for the other context's pair once left the command processor serving neither.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _FRESH % root], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fresh ok" in out.stdout
