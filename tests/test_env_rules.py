"""Known-answer boards and properties of the m,n,k-game rules (CPU).

The reference's environments are empty submodules (.gitmodules:1-9), so no upstream fixture pins
them: parity unpinned.  What pins the rules here are the call-site contract (mcts.hpp:251-271,427,
generate.cpp:75,103: step -> (state, reward[2] by absolute player, done); possible_actions = empty
cells; get_board = planes {player-0 stones, player-1 stones, side to move}; player 0 moves first)
and the standard rules (K or more in a row wins, a full board is a draw {0,0}), written out below as
hand-made positions and as an independent brute-force checker.  Both CPU implementations are held to
them: the oracle (oracle/az_oracle.cpp: azo_env_*) and the library's host twin of the device kernel
(alphazero_amd/csrc/az_rules.h via az_host_env_step); the device kernel is compared with the oracle in
tests/test_gpu_tree.py::test_env_random_playouts_bit_exact.
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import azo
from alphazero_amd import capi


def _oracle_step(N, K, cells, tm, a):
    return azo.env_step(N, K, cells, tm, a)


def _host_step(N, K, cells, tm, a):
    return capi.host_env_step(N, K, cells, tm, a)


IMPLS = [pytest.param(_oracle_step, id="oracle"), pytest.param(_host_step, id="host-twin")]


def play(step, N, K, moves):
    """-> (cells, to_move, reward, done) after the moves ((y, x) pairs), asserting that none but the last ends the game."""
    cells, tm = azo.env_reset(N)
    rew, done = np.zeros(2, np.float32), 0
    for i, (y, x) in enumerate(moves):
        assert done == 0, "move %d after the end" % i
        cells, tm, rew, done = step(N, K, cells, tm, y * N + x)
        assert done >= 0, "move %d (%d,%d) rejected" % (i, y, x)
    return cells, tm, rew, done


def line_winner(N, K, cells):
    """Independent brute force: 1 or 2 if that player has K or more in a row anywhere, else 0."""
    b = np.asarray(cells).reshape(N, N)
    for y in range(N):
        for x in range(N):
            p = b[y, x]
            if p == 0:
                continue
            for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
                yy, xx, n = y, x, 0
                while 0 <= yy < N and 0 <= xx < N and b[yy, xx] == p:
                    n += 1
                    yy += dy
                    xx += dx
                if n >= K:
                    return int(p)
    return 0


@pytest.mark.parametrize("step", IMPLS)
def test_tictactoe_known_answers(step):
    N = K = 3
    # player 0 takes the top row: 0 at (0,0),(0,1),(0,2); player 1 at (1,0),(1,1)
    cells, tm, rew, done = play(step, N, K, [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2)])
    assert done == 1 and list(rew) == [1.0, -1.0]
    assert list(cells) == [1, 1, 1, 2, 2, 0, 0, 0, 0]
    # player 1 wins the anti-diagonal
    cells, tm, rew, done = play(step, N, K, [(0, 0), (0, 2), (0, 1), (1, 1), (2, 2), (2, 0)])
    assert done == 1 and list(rew) == [-1.0, 1.0]
    # a column and the main diagonal
    cells, tm, rew, done = play(step, N, K, [(0, 1), (0, 0), (1, 1), (2, 2), (2, 1)])
    assert done == 1 and list(rew) == [1.0, -1.0]
    cells, tm, rew, done = play(step, N, K, [(0, 0), (0, 1), (1, 1), (0, 2), (2, 2)])
    assert done == 1 and list(rew) == [1.0, -1.0]
    # full board, nobody wins: draw, reward {0, 0}
    draw = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 0), (1, 2), (2, 1), (2, 0), (2, 2)]
    cells, tm, rew, done = play(step, N, K, draw)
    assert done == 1 and list(rew) == [0.0, 0.0] and np.count_nonzero(cells) == 9 and line_winner(N, K, cells) == 0
    # the ninth move completes a line: a win, not a draw
    last = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 0), (1, 2), (2, 1), (2, 2), (2, 0)]
    cells, tm, rew, done = play(step, N, K, last)
    assert done == 1 and list(rew) == [1.0, -1.0]


@pytest.mark.parametrize("step", IMPLS)
def test_first_player_side_to_move_and_planes(step):
    N, K = 5, 4
    cells, tm = azo.env_reset(N)
    assert tm == 0 and not cells.any()
    planes = azo.env_planes(N, cells, tm)
    assert not planes.any()                                 # empty board, player 0 to move: plane 2 is all zero
    cells, tm, rew, done = step(N, K, cells, tm, 2 * N + 3)
    assert tm == 1 and done == 0 and list(rew) == [0.0, 0.0] and cells[2 * N + 3] == 1
    cells, tm, rew, done = step(N, K, cells, tm, 0)
    assert tm == 0 and cells[0] == 2
    planes = azo.env_planes(N, cells, 1)
    assert planes[0, 2, 3] == 1 and planes[0].sum() == 1    # plane 0: player-0 stones
    assert planes[1, 0, 0] == 1 and planes[1].sum() == 1    # plane 1: player-1 stones
    assert planes[2].min() == 1                             # plane 2: side to move, broadcast
    legal = azo.env_legal(N, cells)
    assert len(legal) == N * N - 2 and 0 not in legal and 2 * N + 3 not in legal and list(legal) == sorted(legal)


@pytest.mark.parametrize("step", IMPLS)
def test_gomoku_lines(step):
    N, K = 9, 5
    filler = [(8, 0), (8, 2), (8, 4), (8, 6), (7, 1)]       # player 1's moves: never a line
    # four in a row is not a win, the fifth stone is
    row = [(4, 2), (4, 3), (4, 4), (4, 5)]
    moves = [m for pair in zip(row, filler) for m in pair]
    cells, tm, rew, done = play(step, N, K, moves)
    assert done == 0 and line_winner(N, K, cells) == 0
    cells, tm, rew, done = step(N, K, cells, tm, 4 * N + 6)
    assert done == 1 and list(rew) == [1.0, -1.0]
    # the stone that fills a gap wins too (lines are found around the last stone, both directions)
    gap = [(2, 2), (3, 3), (5, 5), (6, 6)]
    moves = [m for pair in zip(gap, filler) for m in pair]
    cells, tm, rew, done = play(step, N, K, moves)
    cells, tm, rew, done = step(N, K, cells, tm, 4 * N + 4)
    assert done == 1 and list(rew) == [1.0, -1.0]
    # six in a row (overline) counts: K or more
    six = [(1, 0), (1, 1), (1, 2), (1, 4), (1, 5)]
    moves = [m for pair in zip(six, filler) for m in pair]
    cells, tm, rew, done = play(step, N, K, moves)
    cells, tm, rew, done = step(N, K, cells, tm, 1 * N + 3)
    assert done == 1 and line_winner(N, K, cells) == 1
    # a row does not wrap round the edge: (0,7),(0,8),(1,0),(1,1),(1,2) are consecutive cells in memory only
    wrap = [(0, 7), (0, 8), (1, 0), (1, 1), (1, 2)]
    moves = [m for pair in zip(wrap, filler) for m in pair][:-1]
    cells, tm, rew, done = play(step, N, K, moves)
    assert done == 0 and line_winner(N, K, cells) == 0
    # an opponent's stone breaks a line
    broken = [(6, 1), (6, 2), (6, 4), (6, 5)]
    moves = [m for pair in zip(broken, [(6, 3), (8, 2), (8, 4), (8, 6)]) for m in pair]
    cells, tm, rew, done = play(step, N, K, moves)
    cells, tm, rew, done = step(N, K, cells, tm, 6 * N + 6)
    assert done == 0
    # player 1 wins on an anti-diagonal: reward by absolute player
    p0 = [(0, 0), (0, 2), (0, 4), (0, 6), (8, 8)]
    p1 = [(2, 6), (3, 5), (4, 4), (5, 3), (6, 2)]
    moves = [m for pair in zip(p0, p1) for m in pair]
    cells, tm, rew, done = play(step, N, K, moves)
    assert done == 1 and list(rew) == [-1.0, 1.0]


@pytest.mark.parametrize("step", IMPLS)
def test_illegal_moves_are_rejected_and_change_nothing(step):
    N, K = 3, 3
    cells, tm = azo.env_reset(N)
    cells, tm, rew, done = step(N, K, cells, tm, 4)
    for bad in (4, -1, 9, 100):
        c2, t2, r2, d2 = step(N, K, cells, tm, bad)
        assert d2 < 0 and np.array_equal(c2, cells) and t2 == tm


@settings(max_examples=60, deadline=None)
@given(st.sampled_from([(3, 3), (4, 3), (6, 4), (9, 5), (15, 5)]), st.integers(0, 2 ** 31 - 1))
def test_random_playouts_against_the_brute_force_checker(shape, seed):
    """Every step of a random game: stones = moves, sides alternate from player 0, the game ends exactly
    when the brute-force checker sees a line or the board is full, rewards are +1/-1 for winner/loser by
    absolute player and 0/0 for a draw; oracle and host twin agree bit for bit."""
    N, K = shape
    rng = np.random.default_rng(seed)
    ca, ta = azo.env_reset(N)
    cb, tb = ca.copy(), 0
    for ply in range(N * N):
        legal = azo.env_legal(N, ca)
        assert len(legal) == N * N - ply and all(ca[a] == 0 for a in legal)
        a = int(rng.choice(legal))
        mover = ta
        ca, ta, ra, da = azo.env_step(N, K, ca, ta, a)
        cb, tb, rb, db = capi.host_env_step(N, K, cb, tb, a)
        assert np.array_equal(ca, cb) and ta == tb and np.array_equal(ra, rb) and da == db
        assert ca[a] == mover + 1 and np.count_nonzero(ca) == ply + 1 and ta == 1 - mover
        w = line_winner(N, K, ca)
        assert w in (0, mover + 1)
        if w:
            assert da == 1 and ra[mover] == 1.0 and ra[1 - mover] == -1.0
            break
        if ply + 1 == N * N:
            assert da == 1 and list(ra) == [0.0, 0.0]
            break
        assert da == 0 and list(ra) == [0.0, 0.0]
