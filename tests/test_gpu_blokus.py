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


def test_search_matches_the_oracle_bit_for_bit():
    """MCTS over Blokus (four players, irregular action sets, skipped players and terminal leaves inside the
    tree) with the integer-hash evaluator and explicit Dirichlet noise: root visit counts, values and priors of
    every game identical to the oracle's search from the same position.  Positions: the opening, the middle
    game, and late positions where players are skipped and games end inside the search."""
    G, budget, seed = 8, 40, 5
    depth = [0, 1, 7, 20, 41, 52, 58, 60]                  # pieces already on the board, per game
    dev = capi.Blokus(G, seed=seed)
    dev.search_config(vl=3.0, c_puct=3.0, max_budget=budget)
    ref = [azo.Blokus() for _ in range(G)]
    for g in range(G):                                     # the same random lines on both sides
        for mv in range(depth[g]):
            if ref[g].over:
                break
            a, _, _ = ref[g].random_move(seed, g, mv)
            acts = np.full(G, -1, np.int32)
            acts[g] = a
            dev.step(acts)
    _, count = dev.legal_mask(want_mask=False)
    noise = np.zeros((G, int(count.max())), np.float32)
    for g in range(G):
        if count[g]:
            noise[g, :count[g]] = azo.dirichlet(300 + g, int(count[g]))[0]
    dev.set_noise(noise)
    dev.search(budget)
    st = dev.search_stats()
    assert st["restarts"] == 0 and st["max_nodes"] <= budget + 1
    live = 0
    for g in range(G):
        got = dev.search_result(g)
        if ref[g].over:
            assert len(got["actions"]) == 0
            continue
        live += 1
        want = azo.blokus_search(ref[g], budget, noise[g, :count[g]])
        assert got["root_n"] == want["root_n"] == budget + 1
        for k in ("actions", "n", "v", "p"):
            assert np.array_equal(got[k], want[k]), (g, k)
        assert got["n"].sum() == budget
    assert live >= 6
    # a second search from the same positions starts from a fresh root: same answer
    dev.search(budget)
    for g in (0, 3):
        again = dev.search_result(g)
        want = azo.blokus_search(ref[g], budget, noise[g, :count[g]])
        assert np.array_equal(again["n"], want["n"]) and np.array_equal(again["v"], want["v"])


@pytest.mark.parametrize("leaves", [2, 5])
def test_search_with_several_descents_per_round_matches_the_oracle(leaves):
    """az_blokus_search_leaves: the reference's searcher threads on one tree (MCTS(nthreads, ...), mcts.hpp:54-72) as
    `leaves` descents per game and round, virtual loss between them, a descent that runs into a leaf the round is
    already waiting on taken back.  Root statistics of every game against the oracle's rounds, bit for bit: fresh
    roots with noise at several depths of the game (skipped players, terminal leaves), then with tree reuse across a
    move; the visit total is the budget whatever the number of rounds."""
    G, budget, seed = 8, 41, 5                             # 41: the last round issues fewer descents than `leaves`
    depth = [0, 1, 7, 20, 41, 52, 58, 60]
    dev = capi.Blokus(G, seed=seed)
    dev.search_config(vl=3.0, c_puct=3.0, max_budget=budget, reuse=True, leaves=leaves)
    assert capi.load().az_blokus_eval_slots(dev.h) == G * leaves
    ref = [azo.Blokus() for _ in range(G)]
    for g in range(G):
        for mv in range(depth[g]):
            if ref[g].over:
                break
            a, _, _ = ref[g].random_move(seed, g, mv)
            acts = np.full(G, -1, np.int32)
            acts[g] = a
            dev.step(acts)
    _, count = dev.legal_mask(want_mask=False)
    noise = np.zeros((G, int(count.max())), np.float32)
    for g in range(G):
        if count[g]:
            noise[g, :count[g]] = azo.dirichlet(300 + g, int(count[g]))[0]
    dev.set_noise(noise)
    agents = [azo.BlokusMCTS() for _ in range(G)]
    azo.blokus_set_leaves(leaves)
    try:
        def check(tag, with_noise):
            live = 0
            for g in range(G):
                got = dev.search_result(g)
                if ref[g].over:
                    continue
                live += 1
                want = agents[g].search(ref[g], budget, noise[g, :count[g]] if with_noise else None)
                assert got["root_n"] == want["root_n"], (tag, g, got["root_n"], want["root_n"])
                for k in ("actions", "n", "v", "p"):
                    assert np.array_equal(got[k], want[k]), (tag, g, k)
            return live
        dev.search(budget)
        assert check("fresh", True) >= 6
        for g in range(G):
            if not ref[g].over:
                assert dev.search_result(g)["n"].sum() == budget
        # a move (the most visited child), then a search on the kept subtree
        acts = np.full(G, -1, np.int32)
        for g in range(G):
            if ref[g].over:
                continue
            r = dev.search_result(g)
            acts[g] = r["actions"][int(np.argmax(r["n"]))]
            ref[g].step(int(acts[g]))
        dev.step(acts)
        dev.set_noise(None)                                # (a child that was never expanded gives a fresh root)
        dev.search(budget)
        check("after a move", False)
        st = dev.search_stats()
        assert st["restarts"] == 0
        # fewer rounds than simulations: several descents per round did happen
        assert st["rounds"] < 2 * (budget + 3)
    finally:
        azo.blokus_set_leaves(1)


def test_oracle_rounds_of_one_descent_are_the_plain_simulations():
    g = azo.Blokus()
    for mv in range(9):
        g.random_move(3, 0, mv)
    a, b = azo.blokus_round_check(g, 60)
    assert np.array_equal(a, b) and a.sum() == 60


def test_search_with_the_tower_and_two_descents_per_round():
    """The tower evaluates G * leaves positions per round; every simulation is accounted for and the root's priors are
    the tower's softmax at the legal actions."""
    from alphazero_amd import synth
    G, budget, leaves = 12, 24, 2
    Cs = synth.NETCONFIG2
    dev = capi.Blokus(G, seed=9)
    dev.search_config(max_budget=budget, leaves=leaves)
    dev.playout(6)
    dev.sync()
    small = capi.Engine(20, 5, G, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1)
    with pytest.raises(capi.AzError, match="evaluates 24 positions"):
        dev.search_tower(small, budget)
    tower = capi.Engine(20, 5, G * leaves, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1)
    tower.load_weights(synth.make_blob(20, 5, Cs, 31, phead_gain=4.0, fc2_gain=2.0, c_out=91, value_dim=4))
    dev.search_tower(tower, budget)
    pol, _ = tower.tower_forward(np.concatenate([dev.boards(), dev.boards()]))
    _, count = dev.legal_mask(want_mask=False)
    for g in range(G):
        r = dev.search_result(g)
        assert r["root_n"] == budget + 1 and r["n"].sum() == budget and len(r["actions"]) == count[g]
        assert np.array_equal(r["p"], pol[g, r["actions"]])
    st = dev.search_stats()
    assert st["restarts"] == 0 and st["rounds"] < budget + 2


def test_search_at_config_size_2048_games_400_sims():
    """BASELINE configs[3]: 2048 concurrent games x 400 simulations (hash evaluator): every game's root holds
    exactly 400 visits spread over its legal actions, no tree outgrew its arena; a sample against the oracle."""
    G, budget = 2048, 400
    dev = capi.Blokus(G, seed=77)
    dev.search_config(max_budget=budget)
    dev.playout(9)                                         # nine random pieces: past the openings
    dev.sync()
    _, count = dev.legal_mask(want_mask=False)
    dev.search(budget)
    st = dev.search_stats()
    print("blokus 2048 x 400:", st, "legal actions per root: mean %.0f max %d" % (count.mean(), count.max()))
    assert st["restarts"] == 0 and st["max_nodes"] <= budget + 1
    for g in range(0, G, 97):
        r = dev.search_result(g)
        assert r["root_n"] == budget + 1 and r["n"].sum() == budget and len(r["actions"]) == count[g]
        assert (np.diff(r["actions"]) > 0).all() and (r["n"] >= 0).all()
    ref = azo.Blokus()
    for mv in range(9):
        ref.random_move(77, 5, mv)
    want = azo.blokus_search(ref, budget)
    got = dev.search_result(5)
    for k in ("actions", "n", "v", "p"):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("precision", ["fp16", "fp16x3"])
def test_search_with_the_tower_as_evaluator(precision):
    """The Blokus search fed by the HIP tower (20x20, 5 planes in, 91 policy channels and 4 values out) instead
    of the hash evaluator, device to device: the root's priors are the tower's softmax at the legal actions
    (bit for bit: same kernels on the same planes), every simulation is accounted for, values stay in range.
    Either operand format of the tower: the default and the compensated one."""
    from alphazero_amd import synth
    G, budget = 16, 24
    Cs = synth.NETCONFIG2
    dev = capi.Blokus(G, seed=9)
    dev.search_config(max_budget=budget)
    dev.playout(6)
    dev.sync()
    tower = capi.Engine(20, 5, G, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1,
                        precision={"fp16": capi.AZ_PREC_FP16, "fp16x3": capi.AZ_PREC_FP16X3}[precision])
    tower.load_weights(synth.make_blob(20, 5, Cs, 31, phead_gain=4.0, fc2_gain=2.0, c_out=91, value_dim=4))
    with pytest.raises(capi.AzError, match="c_in 5"):
        dev.search_tower(capi.Engine(20, 5, G, Cs=Cs, max_budget=1), budget)       # an m,n,k-shaped tower
    dev.search_tower(tower, budget)
    pol, val = tower.tower_forward(dev.boards())
    _, count = dev.legal_mask(want_mask=False)
    for g in range(G):
        r = dev.search_result(g)
        assert r["root_n"] == budget + 1 and r["n"].sum() == budget and len(r["actions"]) == count[g]
        assert np.array_equal(r["p"], pol[g, r["actions"]])
        vis = r["n"] > 0
        assert (np.abs(r["v"][vis] / r["n"][vis]) <= 1.0 + 1e-6).all() and (r["v"][~vis] == 0).all()
    assert dev.search_stats()["restarts"] == 0


def test_config3_search_2048_games_400_sims_with_the_tower():
    """BASELINE configs[3] as bench.py --config 3 runs it: 2048 Blokus games x 400 simulations with the tower
    (10 x 128 channels, 5 planes in, 91 policy channels, 4 values) as evaluator, nothing leaving the device
    between the rounds: every root holds exactly 400 visits over its legal actions, no tree outgrew its arena,
    and for a sample of games the root's priors are the tower's softmax at the legal actions, bit for bit."""
    from alphazero_amd import synth
    G, budget = 2048, 400
    Cs = synth.NETCONFIG2
    dev = capi.Blokus(G, seed=78)
    dev.search_config(max_budget=budget)
    dev.playout(9)
    dev.sync()
    tower = capi.Engine(20, 5, G, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1)
    tower.load_weights(synth.make_blob(20, 5, Cs, 31, phead_gain=4.0, fc2_gain=2.0, c_out=91, value_dim=4))
    dev.search_tower(tower, budget)
    st = dev.search_stats()
    _, count = dev.legal_mask(want_mask=False)
    print("blokus 2048 x 400 with the tower:", st)
    assert st["restarts"] == 0 and st["max_nodes"] <= budget + 1
    sample = list(range(0, G, 131))
    boards = dev.boards()
    pol, _ = tower.tower_forward(boards[sample])
    for k, g in enumerate(sample):
        r = dev.search_result(g)
        assert r["root_n"] == budget + 1 and r["n"].sum() == budget and len(r["actions"]) == count[g]
        assert (np.diff(r["actions"]) > 0).all() and (r["n"] >= 0).all()
        assert np.array_equal(r["p"], pol[k, r["actions"]])
        vis = r["n"] > 0
        assert (np.abs(r["v"][vis] / r["n"][vis]) <= 1.0 + 1e-6).all() and (r["v"][~vis] == 0).all()
    assert tower.stats()["saturated"] == 0


def test_tree_reuse_across_searches_matches_the_oracle():
    """az_blokus_search_reuse = the reference's __make_root + find (mcts.hpp:385-421): a second search of the same
    position keeps accumulating on its root; after a move the subtree under the move played becomes the tree, with
    its statistics; a move that leads to a never-expanded child, or two moves at once that leave the tree, give a
    fresh root; az_blokus_tree_clear drops everything.  Root statistics against the oracle's search object, game by
    game, bit for bit."""
    G, budget = 6, 40
    dev = capi.Blokus(G, seed=31)
    dev.search_config(max_budget=budget, reuse=True)
    dev.playout(5)
    dev.sync()
    ref, agents = [], []
    for g in range(G):
        o = azo.Blokus()
        for mv in range(5):
            o.random_move(31, g, mv)
        ref.append(o)
        agents.append(azo.BlokusMCTS())

    def check(tag, want_reused=None):
        for g in range(G):
            if ref[g].over:
                continue
            want = agents[g].search(ref[g], budget)
            got = dev.search_result(g)
            assert got["root_n"] == want["root_n"], (tag, g, got["root_n"], want["root_n"], want["reused"])
            for k in ("actions", "n", "v", "p"):
                assert np.array_equal(got[k], want[k]), (tag, g, k)
            if want_reused is not None:
                assert want["reused"] == want_reused[g] if isinstance(want_reused, list) else want["reused"] == want_reused, (tag, g, want["reused"])
    dev.search(budget)
    check("fresh", 0)
    dev.search(budget)                                    # target == root: accumulates (2 x budget + 1)
    check("same root", 2)
    # play: game 0..3 the most visited child (expanded: reused), game 4 the LEAST visited legal child (a stub with
    # n = 0 exists at 40 sims over ~300 children: fresh root), game 5 stays where it is
    acts = np.full(G, -1, np.int32)
    how = []
    for g in range(G):
        r = dev.search_result(g)
        if g < 4:
            k = int(np.argmax(r["n"]))
        elif g == 4:
            k = int(np.argmin(r["n"]))
            assert r["n"][k] == 0
        else:
            how.append(2)
            continue
        acts[g] = r["actions"][k]
        ref[g].step(int(acts[g]))
        how.append(1 if g < 4 else 0)
    dev.step(acts)
    dev.search(budget)
    check("after a move", how)
    # two more moves straight away on game 0 (the second one leaves what was searched: most likely a stub)
    for _ in range(2):
        acts = np.full(G, -1, np.int32)
        r = dev.search_result(0)
        acts[0] = r["actions"][int(np.argmax(r["n"]))]
        ref[0].step(int(acts[0]))
        dev.step(acts)
        acts[0] = -1
        m, _c = dev.legal_mask()
        acts[0] = int(np.flatnonzero(m[0])[-1])            # the last legal action: rarely visited
        ref[0].step(int(acts[0]))
        dev.step(acts)
        dev.search(budget)
        check("two plies on")
    dev.tree_clear()
    for a in agents:
        a.clear()
    dev.search(budget)
    check("after clear", 0)
    assert dev.search_stats()["restarts"] == 0


def test_selfplay_with_tree_reuse_and_warm_up_matches_the_oracle_loop():
    """The reference's generate loop in full for Blokus (az_blokus_search_reuse): a warm-up search at every game
    start (generate.cpp:77), the tree carried from move to move, cleared when the game ends (generate.cpp:113).
    3 games x 100 steps with the hash evaluator: every finished row equals the oracle's loop, game by game."""
    G, budget, n_steps, seed = 3, 6, 100, 12
    dev = capi.Blokus(G, seed=seed)
    dev.search_config(max_budget=budget, reuse=True)
    dev.selfplay_config(ring_rows=G * 2 * 84, max_budget=budget)
    assert dev.K == 4 * budget
    dev.selfplay_moves(budget, n_steps)
    st = dev.selfplay_stats()
    got = dev.replay_drain(G * 2 * 84)
    assert st["dropped"] == 0

    def dev_row(r):
        c = got["counts"][r]
        return (got["planes"][r].tobytes(), int(got["to_move"][r]), got["actions"][r, :c].tobytes(), got["probs"][r, :c].tobytes(),
                got["reward"][r].tobytes())
    dev_games = sorted(_game_blocks(got["planes"], dev_row))
    ref_games, ref_moves, ref_finished = [], 0, 0
    for g in range(G):
        rows, moves, games = azo.blokus_selfplay(n_steps, budget, seed, g, reuse=True)
        ref_moves += moves
        ref_finished += games
        ref_games += _game_blocks([r["planes"] for r in rows],
                                  lambda r, rows=rows: (rows[r]["planes"].tobytes(), int(rows[r]["to_move"]), rows[r]["actions"].astype(np.int32).tobytes(),
                                                        rows[r]["probs"].tobytes(), rows[r]["reward"].tobytes()))
    assert st["moves"] == ref_moves and st["games"] == ref_finished and ref_finished >= 3
    assert dev.search_stats()["restarts"] == 0            # (a tree that outgrew its arena would have restarted: the oracle has no arena)
    assert dev_games == sorted(ref_games)
    # the first row of every game comes from a root searched twice: its policy sums to 2 b / (2 b + 1)
    first = got["planes"][:, :4].reshape(len(got["counts"]), -1).sum(1) == 0
    assert first.sum() == st["games"] and np.allclose(got["probs"][first].sum(1), 2 * budget / (2 * budget + 1.0), atol=1e-6)


def _game_blocks(rows_planes, per_row):
    """split rows (oldest first, each game's rows contiguous) into games at the empty-board rows"""
    blocks, cur = [], []
    for r, pl in enumerate(rows_planes):
        if pl[:4].sum() == 0 and cur:
            blocks.append(cur)
            cur = []
        cur.append(per_row(r))
    if cur:
        blocks.append(cur)
    return blocks


@pytest.mark.parametrize("leaves", [1, 3])
def test_selfplay_rows_match_the_oracle_bit_for_bit(leaves):
    """The move loop (search from a fresh root, row, sampled or arg-max move, step, reward hand-over, next game)
    on 3 games x 90 moves with the hash evaluator: every finished row -- planes, player to move, the visited
    children's actions and n_child / n_root, the four-entry reward -- equals the oracle's loop, game by game; also
    with three descents per round on both sides."""
    G, budget, n_moves, seed = 3, 6, 90, 11
    dev = capi.Blokus(G, seed=seed)
    dev.search_config(max_budget=budget, leaves=leaves)
    azo.blokus_set_leaves(leaves)
    try:
        _selfplay_rows_check(dev, G, budget, n_moves, seed)
    finally:
        azo.blokus_set_leaves(1)


def _selfplay_rows_check(dev, G, budget, n_moves, seed):
    dev.selfplay_config(ring_rows=G * 2 * 84, max_budget=budget)
    dev.selfplay_moves(budget, n_moves)
    st = dev.selfplay_stats()
    got = dev.replay_drain(G * 2 * 84)
    assert dev.selfplay_stats()["rows"] == 0 and st["dropped"] == 0 and st["moves"] == G * n_moves

    def dev_row(r):
        c = got["counts"][r]
        return (got["planes"][r].tobytes(), int(got["to_move"][r]), got["actions"][r, :c].tobytes(), got["probs"][r, :c].tobytes(),
                got["reward"][r].tobytes())
    dev_games = sorted(_game_blocks(got["planes"], dev_row))
    ref_games, ref_finished = [], 0
    for g in range(G):
        rows, moves, games = azo.blokus_selfplay(n_moves, budget, seed, g)
        assert moves == n_moves
        ref_finished += games
        ref_games += _game_blocks([r["planes"] for r in rows],
                                  lambda r, rows=rows: (rows[r]["planes"].tobytes(), int(rows[r]["to_move"]), rows[r]["actions"].astype(np.int32).tobytes(),
                                                        rows[r]["probs"].tobytes(), rows[r]["reward"].tobytes()))
    assert st["games"] == ref_finished >= G and len(got["counts"]) == sum(len(b) for b in ref_games)
    assert dev_games == sorted(ref_games)
    dense = capi.Blokus.dense_policy(got["actions"], got["probs"], got["counts"])
    assert dense.shape[1:] == (91, 20, 20) and np.allclose(dense.reshape(len(dense), -1).sum(1), budget / (budget + 1.0), atol=1e-6)


def test_selfplay_with_the_tower_and_device_noise():
    """The same loop with the HIP tower as evaluator and Dirichlet noise drawn on the device: games finish, rows are
    well-formed (policy mass n_root - 1 over n_root, legal first rows, rewards in {-1, 0, 1} with a winner)."""
    from alphazero_amd import synth
    G, budget = 8, 12
    Cs = [128, 128, 128]
    tower = capi.Engine(20, 5, G, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1)
    tower.load_weights(synth.make_blob(20, 5, Cs, 5, c_out=91, value_dim=4))
    dev = capi.Blokus(G, seed=5)
    dev.search_config(max_budget=budget)
    dev.noise_device(True)
    dev.selfplay_config(ring_rows=G * 84, max_budget=budget)
    dev.selfplay_moves(budget, 84, tower_engine=tower)          # a game has at most 84 moves
    st = dev.selfplay_stats()
    assert st["games"] >= G and st["moves"] == G * 84
    got = dev.replay_drain(G * 84)
    n = len(got["counts"])
    assert n == st["rows"] + 0 and n >= G * 20
    assert np.allclose(got["probs"].sum(1), budget / (budget + 1.0), atol=1e-6)
    assert set(np.unique(got["reward"])) <= {-1.0, 0.0, 1.0} and (got["reward"].max(1) == 1.0).all()
    first = got["planes"][:, :4].reshape(n, -1).sum(1) == 0
    assert first.sum() == st["games"] and (got["to_move"][first] == 0).all()


def test_search_survives_an_arena_too_small_for_the_position():
    """edge_cap far below what 12 simulations of a mid-game position need: the search restarts once from a fresh
    root, then ends with the simulations it has instead of restarting for ever; self-play carries on."""
    G, budget = 4, 12
    dev = capi.Blokus(G, seed=9)
    with pytest.raises(capi.AzError, match="cannot hold a root"):
        dev.search_config(max_budget=budget, edge_cap=600)     # below the floor: a root alone can need more
    dev.search_config(max_budget=budget, edge_cap=1500)
    dev.selfplay_config(ring_rows=G * 84, max_budget=budget)
    dev.selfplay_moves(budget, 40)
    st, ss = dev.selfplay_stats(), dev.search_stats()
    assert st["moves"] == G * 40 and ss["restarts"] > 0 and ss["max_slots"] <= 1500
    for g in range(G):
        r = dev.search_result(g)
        assert 1 <= r["root_n"] <= budget + 1 and int(r["n"].sum()) == r["root_n"] - 1 and (r["n"] >= 0).all()


def test_selfplay_at_config_size_keeps_its_books():
    """2 048 games in lockstep (BASELINE's Blokus size) with a short budget: every game advances one move per
    call, nobody is stuck, finished games hand over exactly their rows, and the books add up."""
    G, budget, n_moves = 2048, 8, 30
    dev = capi.Blokus(G, seed=21)
    dev.search_config(max_budget=budget)
    dev.noise_device(True)
    dev.selfplay_config(ring_rows=G * 30, max_budget=budget)
    dev.selfplay_moves(budget, n_moves)
    st, ss, state = dev.selfplay_stats(), dev.search_stats(), dev.state()
    assert st["moves"] == G * n_moves and st["dropped"] == 0 and ss["restarts"] == 0
    assert not state["over"].any()
    # rows in the ring + moves of the games still in progress = all moves played
    assert st["rows"] + int(state["moves"].sum()) == G * n_moves
    got = dev.replay_drain(G * 30)
    assert len(got["counts"]) == st["rows"]
    if st["rows"]:
        assert (got["counts"] >= 1).all() and (got["counts"] <= budget).all()
        assert np.allclose(got["probs"].sum(1), budget / (budget + 1.0), atol=1e-6)
        assert (got["reward"].max(1) == 1.0).all()


def test_replay_drain_in_pieces_equals_one_drain():
    """Two identical runs (hash evaluator, same seed): draining 7 rows at a time returns the same rows in the same
    order as one big drain, and leaves the ring empty."""
    G, budget, n_moves = 3, 4, 75
    runs = []
    for _ in range(2):
        dev = capi.Blokus(G, seed=33)
        dev.search_config(max_budget=budget)
        dev.selfplay_config(ring_rows=G * 84, max_budget=budget)
        dev.selfplay_moves(budget, n_moves)
        runs.append(dev)
    whole = runs[0].replay_drain(G * 84)
    n = len(whole["counts"])
    assert n >= 50
    parts = []
    while True:
        got = runs[1].replay_drain(7)
        if len(got["counts"]) == 0:
            break
        parts.append(got)
    assert runs[1].selfplay_stats()["rows"] == 0
    for key in ("planes", "to_move", "counts", "actions", "probs", "reward"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key]), key
