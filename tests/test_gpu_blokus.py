"""The Blokus kernels (one game per wavefront, legality by bit boards and ballots) against the CPU oracle,
bit for bit: legal masks, transitions, turn order with skipped players, terminal rewards.  PARITY UNPINNED
against the reference (its Blokus submodule is an empty directory): the oracle is pinned by
tests/test_blokus_rules.py only."""
import numpy as np
import pytest

import azo
from alphazero_amd import capi

pytestmark = pytest.mark.gpu
A = 91 * 400


def oracle_state(games):
    return (np.stack([g.planes() for g in games]), np.array([g.to_move for g in games]),
            np.array([g.over for g in games]), np.stack([g.used() for g in games]))


def test_reset_and_first_moves_match_the_oracle():
    G = 5
    dev = capi.Blokus(G)
    ref = [azo.Blokus() for _ in range(G)]
    mask, count = dev.legal_mask()
    for g in range(G):
        m, n = ref[g].legal()
        assert count[g] == n == 58 and np.array_equal(mask[g], m)
    assert not dev.boards().any()
    with pytest.raises(capi.AzError, match="illegal"):
        dev.step(np.array([5 * 20 + 5] + [-1] * (G - 1), np.int32))      # monomino away from the corner
    with pytest.raises(capi.AzError, match="illegal"):
        dev.step(np.array([A] + [-1] * (G - 1), np.int32))
    assert not dev.boards().any()                                         # refused moves change nothing


def test_step_by_step_games_match_the_oracle_bit_for_bit():
    """Host-chosen actions through az_blokus_step, every game a different random line, -1 for games that are
    left alone in a step; compared after every step: planes, player to move (skips included), pieces-played
    masks, legal count, and the full legal mask every few steps; rewards and done flags at the end."""
    G = 12
    dev = capi.Blokus(G)
    ref = [azo.Blokus() for _ in range(G)]
    rng = np.random.default_rng(3)
    ctr = [0] * G
    for step in range(90):
        acts = np.full(G, -1, np.int32)
        exp_rew, exp_done = np.zeros((G, 4), np.float32), np.zeros(G, np.uint8)
        for g in range(G):
            exp_done[g] = ref[g].over
            if ref[g].over or rng.random() < 0.1:                         # finished, or sits this step out
                continue
            a, rew, done = ref[g].random_move(11, g, ctr[g])
            ctr[g] += 1
            acts[g], exp_rew[g], exp_done[g] = a, rew, done
        rew, done = dev.step(acts)
        assert np.array_equal(rew, exp_rew) and np.array_equal(done, exp_done)
        planes, tm, over, used = oracle_state(ref)
        st = dev.state()
        live = ~over
        assert np.array_equal(st["over"].astype(bool), over) and np.array_equal(st["used"], used)
        assert np.array_equal(st["to_move"][live], tm[live])
        devp = dev.boards()
        assert np.array_equal(devp[:, :4], planes[:, :4]) and np.array_equal(devp[live, 4], planes[live, 4])
        mask, count = dev.legal_mask(want_mask=(step % 7 == 0))
        for g in range(G):
            m, n = ref[g].legal()
            assert count[g] == n
            if mask is not None:
                assert np.array_equal(mask[g], m)
        if over.all():
            break
    assert over.all()
    fr = dev.state()["final_reward"]
    for g in range(G):
        s = ref[g].scores()
        assert fr[g].tolist() == [1.0 if v == s.max() else -1.0 for v in s] or (len(set(s.tolist())) == 1 and not fr[g].any())
    with pytest.raises(capi.AzError, match="illegal"):
        dev.step(np.zeros(G, np.int32))                                   # a finished game takes no move


def test_device_playouts_match_the_oracle_at_config_size():
    """BASELINE configs[3] size: 2048 concurrent games played to the end on the device by uniformly random legal
    moves (counter-based draws), a sample of games replayed by the oracle with the same draws: final boards,
    pieces and rewards identical; every game through size-independent properties."""
    G, seed, base = 2048, 2024, 100
    dev = capi.Blokus(G, seed=seed, game_slot_base=base)
    dev.playout(30)
    dev.sync()
    st = dev.state()
    assert (st["moves"] == 30).all() and not st["over"].any()             # nobody is stuck after 30 pieces
    dev.playout(60)
    dev.sync()
    st = dev.state()
    assert st["over"].all()
    planes = dev.boards()
    squares = planes[:, :4].reshape(G, 4, 400).sum(2)
    assert (planes[:, :4].sum(1) <= 1).all()                              # one owner per cell
    sizes = np.array([1, 2, 3, 3, 4, 4, 4, 4, 4] + [5] * 12)
    used_sq = np.array([[sizes[[k for k in range(21) if (int(u) >> k) & 1]].sum() for u in row] for row in st["used"]])
    assert np.array_equal(squares, used_sq)                               # squares on the board = pieces played
    _, count = dev.legal_mask(want_mask=False)
    assert not count.any()
    fr = st["final_reward"]
    score = np.where(used_sq == 89, 15, used_sq - 89)
    for g in range(G):
        best = score[g].max()
        if score[g].min() != best and not (used_sq[g] == 89).any():
            assert fr[g].tolist() == [1.0 if v == best else -1.0 for v in score[g]]
    assert 40 <= st["moves"].min() and st["moves"].max() <= 84
    for g in (0, 1, 777, 2047):                                           # the same games on the oracle
        ref = azo.Blokus()
        mv = 0
        rew = None
        while not ref.over:
            _, rew, _ = ref.random_move(seed, base + g, mv)
            mv += 1
        assert mv == st["moves"][g]
        assert np.array_equal(ref.planes()[:4], planes[g, :4]) and np.array_equal(ref.used(), st["used"][g])
        assert np.array_equal(rew, fr[g])
