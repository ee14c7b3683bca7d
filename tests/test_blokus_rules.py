"""Blokus rules: the stand-in for the upstream pin that cannot exist (the reference's Blokus submodule is an
empty directory, .gitmodules:7-9: PARITY UNPINNED).  Applied to the oracle (oracle/blokus_oracle.cpp) and to
the library's host twin of its device rules (az_blokus_host_*, same header as the kernels):
  * the orientation table against a brute-force enumeration of all fixed polyominoes of 1..5 cells,
  * hand-made known-answer positions (openings, corner contact, edge contact, piece reuse, skipping, scoring),
  * hypothesis playouts against invariants checked from the raw board, and the two implementations against
    each other move by move."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import azo
from alphazero_amd import capi

N, A = 20, 91 * 400
PIECE_SIZES = [1, 2, 3, 3, 4, 4, 4, 4, 4] + [5] * 12


def act(o, y, x):
    return o * 400 + y * 20 + x


def fixed_polyominoes(n):
    """All fixed polyominoes of n cells (translations identified), by growing from one cell."""
    shapes = {((0, 0),)}
    for _ in range(n - 1):
        nxt = set()
        for s in shapes:
            cells = set(s)
            for (y, x) in s:
                for (dy, dx) in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    c = (y + dy, x + dx)
                    if c not in cells:
                        t = cells | {c}
                        my, mx = min(a for a, _ in t), min(b for _, b in t)
                        nxt.add(tuple(sorted((a - my, b - mx) for a, b in t)))
        shapes = nxt
    return shapes


def orient_sets(get):
    out = {}
    for o in range(91):
        piece, h, w, cells = get(o)
        assert h == 1 + max(c[0] for c in cells) and w == 1 + max(c[1] for c in cells)
        assert min(c[0] for c in cells) == 0 and min(c[1] for c in cells) == 0
        out[o] = (piece, tuple(sorted(cells)))
    return out


def test_orientation_table_is_every_fixed_polyomino_once_and_both_tables_agree():
    a, b = orient_sets(azo.Blokus.orient), orient_sets(capi.blokus_orient)
    assert a == b                                             # same order: the action index means the same
    assert [p for p, _ in a.values()] == sorted(p for p, _ in a.values())     # grouped by piece, pieces in order
    per_piece = [sum(1 for p, _ in a.values() if p == k) for k in range(21)]
    assert per_piece == [1, 2, 2, 4, 2, 8, 4, 1, 4, 8, 2, 8, 8, 8, 4, 4, 4, 4, 1, 8, 4]
    assert [len(c) for _, c in a.values()] == [PIECE_SIZES[p] for p, _ in a.values()]
    shapes = [c for _, c in a.values()]
    assert len(set(shapes)) == 91
    brute = set().union(*(fixed_polyominoes(n) for n in range(1, 6)))
    assert len(brute) == 1 + 2 + 6 + 19 + 63 and set(shapes) == brute
    # orientations of one piece are images of each other under the 8 symmetries
    def canon(cells):
        best = None
        for t in range(8):
            pts = [(y, -x) if t >= 4 else (y, x) for y, x in cells]
            for _ in range(t & 3):
                pts = [(x, -y) for y, x in pts]
            my, mx = min(p[0] for p in pts), min(p[1] for p in pts)
            k = tuple(sorted((p[0] - my, p[1] - mx) for p in pts))
            best = k if best is None or k < best else best
        return best
    free = {}
    for p, c in a.values():
        free.setdefault(p, set()).add(canon(c))
    assert all(len(v) == 1 for v in free.values()) and len({next(iter(v)) for v in free.values()}) == 21


def both():
    return [("oracle", azo.Blokus()), ("host twin", HostAdapter())]


class HostAdapter:
    """the library's host twin behind the oracle's method names"""

    def __init__(self):
        self.g = capi.BlokusHost()

    def step(self, a):
        try:
            return self.g.step(a)
        except capi.AzError:
            return np.zeros(4, np.float32), -1

    def legal(self):
        return self.g.legal()

    def cells(self):
        return self.g.info()["cells"]

    def scores(self):
        return self.g.info()["scores"]

    def used(self):
        return self.g.info()["used"]

    to_move = property(lambda self: self.g.info()["to_move"])
    over = property(lambda self: self.g.info()["over"])


O_I1 = 0          # the monomino
O_I2_H = 1        # domino, horizontal
O_I5_H = [o for o in range(91) if azo.Blokus.orient(o)[3] == [(0, k) for k in range(5)]][0]
O_I5_V = [o for o in range(91) if azo.Blokus.orient(o)[3] == [(k, 0) for k in range(5)]][0]
O_O4 = [o for o in range(91) if azo.Blokus.orient(o)[3] == [(0, 0), (0, 1), (1, 0), (1, 1)]][0]


@pytest.mark.parametrize("which", [0, 1], ids=["oracle", "host-twin"])
def test_known_answer_positions(which):
    g = both()[which][1]
    mask, n = g.legal()
    # opening of player 0: exactly the placements that cover (0,0): one per cell of every orientation whose
    # translate keeps the piece on the board = sum over orientations of (cells that can sit on the corner)
    expected = 0
    for o in range(91):
        _, h, w, cells = azo.Blokus.orient(o)
        expected += sum(1 for (cy, cx) in cells if cy == 0 and cx == 0)      # the anchor is the bounding-box corner
    assert n == expected == 58
    assert mask[act(O_I1, 0, 0)] and not mask[act(O_I1, 0, 1)] and not mask[act(O_I1, 19, 19)]
    assert g.step(act(O_I1, 5, 5))[1] == -1                      # not on the corner
    assert g.step(act(O_I5_H, 0, 16))[1] == -1                   # on the board but the wrong corner for player 0
    assert g.step(A)[1] == -1 and g.step(-1)[1] == -1            # out of range
    rew, done = g.step(act(O_I5_H, 0, 0))                        # player 0: I5 along the top edge
    assert done == 0 and not rew.any() and g.to_move == 1
    assert g.step(act(O_I5_H, 0, 0))[1] == -1                    # occupied / not player 1's corner
    assert g.step(act(O_I5_V, 0, 19))[1] == 0 and g.to_move == 2  # player 1: I5 down the right edge
    assert g.step(act(O_O4, 18, 18))[1] == 0 and g.to_move == 3   # player 2: square in the bottom-right corner
    assert g.step(act(O_I1, 19, 0))[1] == 0 and g.to_move == 0    # player 3: monomino
    c = g.cells().reshape(20, 20)
    assert (c[0, :5] == 1).all() and (c[:5, 19] == 2).all() and (c[18:, 18:] == 3).all() and c[19, 0] == 4
    assert int((c != 0).sum()) == 5 + 5 + 4 + 1
    mask, n = g.legal()                                          # player 0's second piece
    assert not mask[act(O_I5_H, 0, 0)] and not mask[act(O_I5_V, 1, 0)]       # I5 is spent (any orientation)
    assert not mask[act(O_I1, 1, 0)] and not mask[act(O_I1, 0, 5)]           # edge contact with own squares
    assert mask[act(O_I1, 1, 5)]                                             # corner contact at (1,5)
    assert not mask[act(O_I1, 1, 6)] and not mask[act(O_I1, 10, 10)]         # no contact at all
    assert mask[act(O_I2_H, 1, 5)] and not mask[act(O_I2_H, 1, 4)]           # (1,4) would touch (0,4) by an edge
    # every legal action: contact only at (1,5) [the other diagonal neighbours of the bar are off the board]
    for a in np.flatnonzero(mask)[:200]:
        _, _, _, cells = azo.Blokus.orient(a // 400)
        y0, x0 = (a % 400) // 20, a % 20
        assert (1, 5) in {(y0 + cy, x0 + cx) for cy, cx in cells}
    # opponents' squares may be touched by an edge: player 0 next to player 1's bar is fine if own contact holds
    assert g.step(act(O_I1, 1, 5))[1] == 0
    assert g.used()[0] == (1 << 10) | 1 and g.used()[3] == 1                 # I5 is piece 10, I1 piece 0
    assert g.scores().tolist() == [-(89 - 6), -(89 - 5), -(89 - 4), -(89 - 1)]


@pytest.mark.parametrize("which", [0, 1], ids=["oracle", "host-twin"])
def test_a_finished_game_scores_and_rewards(which):
    g = both()[which][1]
    ref = azo.Blokus()
    mv = 0
    last = None
    while not ref.over:
        a, rew_ref, done_ref = ref.random_move(7, 3, mv)
        rew, done = g.step(a)
        assert done == done_ref and np.array_equal(rew, rew_ref)
        last = rew
        mv += 1
    s = g.scores()
    left = [sum(PIECE_SIZES[k] for k in range(21) if not (int(g.used()[p]) >> k) & 1) for p in range(4)]
    assert s.tolist() == [-x if x else None for x in left] or all(x > 0 for x in left) and s.tolist() == [-x for x in left]
    best = s.max()
    assert last.tolist() == [1.0 if v == best else -1.0 for v in s] or (len(set(s.tolist())) == 1 and not last.any())
    assert g.legal()[1] == 0 and g.step(int(0))[1] == -1          # nothing more to play
    assert 40 <= mv <= 84


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 2 ** 32 - 1), stream=st.integers(0, 1000))
def test_random_playouts_hold_the_rules_and_both_implementations_agree(seed, stream):
    ref, twin = azo.Blokus(), HostAdapter()
    board = np.zeros((20, 20), np.int64)
    placed = [0, 0, 0, 0]
    mv = 0
    while not ref.over:
        p = ref.to_move
        assert twin.to_move == p
        if mv % 9 == 0:                                           # the full masks agree (costly on the twin)
            m1, n1 = ref.legal()
            m2, n2 = twin.legal()
            assert n1 == n2 and np.array_equal(m1, m2)
        a, rew, done = ref.random_move(seed, stream, mv)
        rew2, done2 = twin.step(a)
        assert done2 == done and np.array_equal(rew, rew2)
        # the move obeys the rules, judged from the raw board kept here
        _, _, _, cells = azo.Blokus.orient(a // 400)
        y0, x0 = (a % 400) // 20, a % 20
        sq = [(y0 + cy, x0 + cx) for cy, cx in cells]
        assert all(0 <= y < 20 and 0 <= x < 20 and board[y, x] == 0 for y, x in sq)
        edge = [(y + dy, x + dx) for y, x in sq for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1))]
        assert not any(0 <= y < 20 and 0 <= x < 20 and board[y, x] == p + 1 for y, x in edge)
        diag = [(y + dy, x + dx) for y, x in sq for dy, dx in ((1, 1), (1, -1), (-1, 1), (-1, -1))]
        if placed[p] == 0:
            assert [(0, 0), (0, 19), (19, 19), (19, 0)][p] in sq
        else:
            assert any(0 <= y < 20 and 0 <= x < 20 and board[y, x] == p + 1 for y, x in diag)
        for y, x in sq:
            board[y, x] = p + 1
        placed[p] += 1
        mv += 1
        # players passed over on the way to the next mover really have no move (checked from the raw board)
        if not done and stream % 4 == 0:
            q = (p + 1) % 4
            while q != ref.to_move:
                assert not any_move_exists(board, q, int(ref.used()[q]))
                q = (q + 1) % 4
    assert np.array_equal(ref.cells().reshape(20, 20), board) and np.array_equal(twin.cells().reshape(20, 20), board)
    assert np.array_equal(ref.scores(), twin.scores()) and np.array_equal(ref.used(), twin.used())
    # the game ended because nobody can move: check it from scratch for every player on the final board
    for p in range(4):
        assert not any_move_exists(board, p, int(ref.used()[p]))


def any_move_exists(board, p, used):
    own = board == p + 1
    for o in range(91):
        piece, h, w, cells = azo.Blokus.orient(o)
        if used >> piece & 1:
            continue
        for y0 in range(20 - h + 1):
            for x0 in range(20 - w + 1):
                sq = [(y0 + cy, x0 + cx) for cy, cx in cells]
                if any(board[y, x] != 0 for y, x in sq):
                    continue
                if any(0 <= y + dy < 20 and 0 <= x + dx < 20 and own[y + dy, x + dx]
                       for y, x in sq for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1))):
                    continue
                if used == 0:
                    if [(0, 0), (0, 19), (19, 19), (19, 0)][p] in sq:
                        return True
                elif any(0 <= y + dy < 20 and 0 <= x + dx < 20 and own[y + dy, x + dx]
                         for y, x in sq for dy, dx in ((1, 1), (1, -1), (-1, 1), (-1, -1))):
                    return True
    return False


def test_oracle_search_object_keeps_its_tree_like_the_reference():
    """azo.BlokusMCTS restates __make_root + find (mcts.hpp:385-421) for the Blokus Env: a first search equals the
    fresh-root search; a second search of the same position keeps accumulating on the root (what generate.cpp:77's
    warm-up does to the first move: n = 1 + 2 b); after a move the kept child's visits carry over; clear() drops
    the tree."""
    budget = 30
    g = azo.Blokus()
    for mv in range(7):
        g.random_move(8, 0, mv)
    m = azo.BlokusMCTS()
    a, b = azo.blokus_search(g, budget), m.search(g, budget)
    assert b["reused"] == 0 and all(np.array_equal(a[k], b[k]) for k in ("actions", "n", "v", "p")) and b["root_n"] == budget + 1
    c = m.search(g, budget)
    assert c["reused"] == 2 and c["root_n"] == 2 * budget + 1 and int(c["n"].sum()) == 2 * budget
    k = int(np.argmax(c["n"]))
    kept = int(c["n"][k])
    g.step(int(c["actions"][k]))
    d = m.search(g, budget)
    assert d["reused"] == 1 and d["root_n"] == kept + budget and int(d["n"].sum()) == kept + budget - 1
    assert (d["p"] > 0).all()                              # priors of the kept node: the evaluator's, no noise mixed in
    m.clear()
    e = m.search(g, budget)
    assert e["reused"] == 0 and e["root_n"] == budget + 1
    f = azo.blokus_search(g, budget)
    assert all(np.array_equal(e[k], f[k]) for k in ("actions", "n", "v", "p"))
