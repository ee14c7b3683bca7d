"""ctypes bindings of oracle/liboracle.so (CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the alphazero_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float))


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.azo_table_create.restype = C.c_void_p
        L.azo_tower_create.restype = C.c_void_p
        L.azo_mcts_create.restype = C.c_void_p
        L.azo_tower_blob_size.restype = C.c_size_t
        L.azo_mcts_eval_count.restype = C.c_long
        L.azo_philox_u32.restype = C.c_uint32
        L.azo_blokus_new.restype = C.c_void_p
        L.azo_philox_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def env_reset(N):
    cells = np.zeros(N * N, np.uint8)
    return cells, 0


def env_step(N, K, cells, to_move, action):
    cells = cells.copy()
    tm = C.c_int(to_move)
    rew = np.zeros(2, np.float32)
    done = lib().azo_env_step(N, K, _p(cells, C.c_uint8), C.byref(tm), int(action), _p(rew, C.c_float))
    return cells, tm.value, rew, done


def env_planes(N, cells, to_move):
    planes = np.zeros(3 * N * N, np.uint8)
    lib().azo_env_planes(N, _p(np.ascontiguousarray(cells), C.c_uint8), int(to_move), _p(planes, C.c_uint8))
    return planes.reshape(3, N, N)


def env_legal(N, cells):
    acts = np.zeros(N * N, np.int32)
    k = lib().azo_env_legal(N, _p(np.ascontiguousarray(cells), C.c_uint8), _p(acts, C.c_int32))
    return acts[:k].copy()


def eval_hash(N, planes):
    pol = np.zeros(N * N, np.float32)
    val = np.zeros(2, np.float32)
    planes = np.ascontiguousarray(planes, np.uint8).reshape(-1)
    lib().azo_eval_hash(None, N, _p(planes, C.c_uint8), _p(pol, C.c_float), _p(val, C.c_float))
    return pol, val


def philox(seed, stream, counter, lane=0):
    return int(lib().azo_philox_u32(seed, stream, counter, lane))


def dirichlet(state, size, alpha=0.05):
    st = C.c_uint64(state)
    out = np.zeros(size, np.float32)
    lib().azo_dirichlet(C.byref(st), size, C.c_float(alpha), _p(out, C.c_float))
    return out, st.value


class Table:
    def __init__(self, N, planes=None, policy=None, value=None):
        self.N = N
        self.h = C.c_void_p(lib().azo_table_create(N))
        if planes is not None:
            for i in range(len(planes)):
                self.add(planes[i], policy[i], value[i])

    def add(self, planes, policy, value):
        planes = np.ascontiguousarray(planes, np.uint8).reshape(-1)
        policy = np.ascontiguousarray(policy, np.float32).reshape(-1)
        value = np.ascontiguousarray(value, np.float32).reshape(-1)
        lib().azo_table_add(self.h, _p(planes, C.c_uint8), _p(policy, C.c_float), _p(value, C.c_float))

    def misses(self):
        return lib().azo_table_misses(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_table_destroy(self.h)
            self.h = None


class Tower:
    def __init__(self, N, c_in, Cs, blob, c_out=1):
        self.N, self.c_in, self.c_out = N, c_in, c_out
        Cs = np.asarray(Cs, np.int32)
        blob = np.ascontiguousarray(blob, np.float32)
        need = lib().azo_tower_blob_size(N, c_in, c_out, _p(Cs, C.c_int), len(Cs))
        assert need == blob.size, (need, blob.size)
        self.h = C.c_void_p(lib().azo_tower_create(N, c_in, c_out, _p(Cs, C.c_int), len(Cs), _p(blob, C.c_float), C.c_size_t(blob.size)))
        assert self.h

    @staticmethod
    def blob_size(N, c_in, Cs, c_out=1):
        Cs = np.asarray(Cs, np.int32)
        return lib().azo_tower_blob_size(N, c_in, c_out, _p(Cs, C.c_int), len(Cs))

    def forward(self, boards):
        boards = np.ascontiguousarray(boards, np.uint8)
        B = boards.shape[0]
        pol = np.zeros((B, self.c_out * self.N * self.N), np.float32)
        val = np.zeros((B, 2), np.float32)
        lib().azo_tower_forward(self.h, _p(boards, C.c_uint8), B, _p(pol, C.c_float), _p(val, C.c_float))
        return pol, val

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_tower_destroy(self.h)
            self.h = None


class MCTS:
    """One reference MCTS<Env> instance at nthreads=1 (mcts.hpp:18-581)."""

    def __init__(self, N, K, vl=3.0, c_puct=3.0, evaluator="hash", user=None):
        L = lib()
        self.N, self.K = N, K
        self._keep = user
        if evaluator == "hash":
            fn, up = C.cast(L.azo_eval_hash, C.c_void_p), None
        elif evaluator == "table":
            fn, up = C.cast(L.azo_eval_table, C.c_void_p), user.h
        elif evaluator == "tower":
            fn, up = C.cast(L.azo_eval_tower, C.c_void_p), user.h
        else:  # python callable (planes[3,N,N]) -> (policy, value)
            def tramp(_u, n, planes, pol, val):
                pl = np.ctypeslib.as_array(planes, shape=(3 * n * n,)).reshape(3, n, n)
                p, v = evaluator(pl)
                np.ctypeslib.as_array(pol, shape=(n * n,))[:] = np.asarray(p, np.float32).reshape(-1)
                np.ctypeslib.as_array(val, shape=(2,))[:] = np.asarray(v, np.float32)
            self._cb = EVAL_FN(tramp)
            fn, up = C.cast(self._cb, C.c_void_p), None
        self.h = C.c_void_p(L.azo_mcts_create(N, K, C.c_float(vl), C.c_float(c_puct), fn, up))

    def search(self, history, budget, noise=None):
        hist = np.ascontiguousarray(history, np.int32)
        out = np.zeros(self.N * self.N, np.float32)
        nz = None if noise is None else _p(np.ascontiguousarray(noise, np.float32), C.c_float)
        rc = lib().azo_mcts_search(self.h, _p(hist, C.c_int32), len(hist), int(budget), nz, _p(out, C.c_float))
        if rc != 0:
            raise ValueError("illegal history")
        return out

    def search_k(self, history, budget, K, noise=None):
        hist = np.ascontiguousarray(history, np.int32)
        out = np.zeros(self.N * self.N, np.float32)
        nz = None if noise is None else _p(np.ascontiguousarray(noise, np.float32), C.c_float)
        rc = lib().azo_mcts_search_k(self.h, _p(hist, C.c_int32), len(hist), int(budget), int(K), nz, _p(out, C.c_float))
        if rc != 0:
            raise ValueError("illegal history")
        return out

    def set_leaves(self, K):
        lib().azo_mcts_set_leaves(self.h, int(K))

    def clear(self):
        lib().azo_mcts_clear(self.h)

    def seed(self, s):
        lib().azo_mcts_seed(self.h, C.c_uint64(s))

    def root_n(self):
        return lib().azo_mcts_root_n(self.h)

    def eval_count(self):
        return lib().azo_mcts_eval_count(self.h)

    def dump(self):
        L = lib()
        n = L.azo_mcts_dump(self.h, 0, None, None, None, None, None, None)
        d = {k: np.zeros(n, np.int32) for k in ("parent", "action", "n", "kind")}
        d["v"] = np.zeros(n, np.float32)
        d["p"] = np.zeros(n, np.float32)
        L.azo_mcts_dump(self.h, n, _p(d["parent"], C.c_int32), _p(d["action"], C.c_int32), _p(d["n"], C.c_int32),
                        _p(d["v"], C.c_float), _p(d["p"], C.c_float), _p(d["kind"], C.c_int32))
        return d

    def selfplay_game(self, budget, seed, slot, noise, avg_len, game_count, sample_counter, max_moves=None):
        N = self.N
        A = N * N
        max_moves = max_moves or A
        cells = np.zeros((max_moves, A), np.uint8)
        tomove = np.zeros(max_moves, np.uint8)
        policy = np.zeros((max_moves, A), np.float32)
        actions = np.zeros(max_moves, np.int32)

        class Res(C.Structure):
            _fields_ = [("n_moves", C.c_int), ("reward", C.c_float * 2)]
        res = Res()
        al = C.c_float(avg_len)
        gc = C.c_int64(game_count)
        sc = C.c_uint32(sample_counter)
        nz = None if noise is None else _p(np.ascontiguousarray(noise, np.float32), C.c_float)
        n = lib().azo_selfplay_game(self.h, int(budget), C.c_uint64(seed), C.c_uint32(slot), nz, C.byref(al), C.byref(gc),
                                    C.byref(sc), max_moves, _p(cells, C.c_uint8), _p(tomove, C.c_uint8),
                                    _p(policy, C.c_float), _p(actions, C.c_int32), C.byref(res))
        assert n > 0
        return dict(n_moves=n, cells=cells[:n], to_move=tomove[:n], policy=policy[:n], actions=actions[:n],
                    reward=np.array([res.reward[0], res.reward[1]], np.float32), avg_len=al.value,
                    game_count=gc.value, sample_counter=sc.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_mcts_destroy(self.h)
            self.h = None


def replay_symmetries(N, state, policy):
    state = np.ascontiguousarray(state, np.uint8)
    Cn = state.shape[0]
    policy = np.ascontiguousarray(policy, np.float32)
    so = np.zeros((8, Cn, N, N), np.uint8)
    po = np.zeros((8, N, N), np.float32)
    lib().azo_replay_symmetries(N, Cn, _p(state, C.c_uint8), _p(policy, C.c_float), _p(so, C.c_uint8), _p(po, C.c_float))
    return so, po


def canonical_tree(d):
    """Tuple form of a BFS dump for exact comparison."""
    return [(int(a), int(b), int(c), float(e), float(f), int(g)) for a, b, c, e, f, g in
            zip(d["parent"], d["action"], d["n"], d["v"], d["p"], d["kind"])]


class Blokus:
    """One Blokus game of the oracle (oracle/blokus_oracle.cpp): 20x20, four players, 91 orientations,
    action = o * 400 + y * 20 + x."""
    N, ORIENTS, ACTIONS, PLAYERS = 20, 91, 91 * 400, 4

    def __init__(self):
        self.h = C.c_void_p(lib().azo_blokus_new())

    @staticmethod
    def orient(o):
        piece, h, w = C.c_int(), C.c_int(), C.c_int()
        cells = np.zeros((5, 2), np.int8)
        n = lib().azo_blokus_orient(int(o), C.byref(piece), C.byref(h), C.byref(w), _p(cells, C.c_int8))
        return piece.value, h.value, w.value, [tuple(int(v) for v in c) for c in cells[:n]]

    def reset(self):
        lib().azo_blokus_reset(self.h)

    def step(self, action):
        rew = np.zeros(4, np.float32)
        done = lib().azo_blokus_step(self.h, int(action), _p(rew, C.c_float))
        return rew, done

    def random_move(self, seed, stream, counter):
        rew = np.zeros(4, np.float32)
        done = C.c_int()
        a = lib().azo_blokus_random_move(self.h, C.c_uint64(seed), C.c_uint32(stream), C.c_uint32(counter), _p(rew, C.c_float), C.byref(done))
        return a, rew, done.value

    def legal(self):
        mask = np.zeros(self.ACTIONS, np.uint8)
        n = lib().azo_blokus_legal(self.h, _p(mask, C.c_uint8))
        return mask, n

    def cells(self):
        c = np.zeros(400, np.uint8)
        lib().azo_blokus_cells(self.h, _p(c, C.c_uint8))
        return c

    def planes(self):
        pl = np.zeros((5, 20, 20), np.uint8)
        lib().azo_blokus_planes(self.h, _p(pl, C.c_uint8))
        return pl

    def used(self):
        u = np.zeros(4, np.uint32)
        lib().azo_blokus_used(self.h, _p(u, C.c_uint32))
        return u

    def scores(self):
        s = np.zeros(4, np.int32)
        lib().azo_blokus_scores(self.h, _p(s, C.c_int32))
        return s

    @property
    def to_move(self):
        return lib().azo_blokus_to_move(self.h)

    @property
    def over(self):
        return bool(lib().azo_blokus_over(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_blokus_free(self.h)
            self.h = None


def blokus_search(game, budget, noise=None, vl=3.0, c_puct=3.0):
    """MCTS of the oracle from the game's current position (fresh root): dict of the root's children."""
    _, k = game.legal()
    out = dict(actions=np.zeros(k, np.int32), n=np.zeros(k, np.int32), v=np.zeros(k, np.float32), p=np.zeros(k, np.float32))
    rn, nodes = C.c_int32(), C.c_int64()
    nz = None if noise is None else _p(np.ascontiguousarray(noise, np.float32), C.c_float)
    got = lib().azo_blokus_search(game.h, int(budget), C.c_float(vl), C.c_float(c_puct), nz, _p(out["actions"], C.c_int32),
                                  _p(out["n"], C.c_int32), _p(out["v"], C.c_float), _p(out["p"], C.c_float), C.byref(rn), C.byref(nodes))
    assert got == k
    out["root_n"], out["nodes"] = rn.value, nodes.value
    return out


def blokus_set_leaves(leaves):
    """descents per round of every Blokus search of the oracle from here on (MCTS(nthreads, ...), mcts.hpp:54-72,
    as lockstep rounds); 1 = one simulation after the other"""
    lib().azo_blokus_set_leaves(int(leaves))


def blokus_round_check(game, budget, vl=3.0, c_puct=3.0):
    """the same fresh-root search through simulate() and through round_k(leaves = 1): both visit-count vectors"""
    k = game.legal()[1]
    a, b = np.zeros(k, np.int32), np.zeros(k, np.int32)
    got = lib().azo_blokus_round_check(game.h, int(budget), C.c_float(vl), C.c_float(c_puct), _p(a, C.c_int32), _p(b, C.c_int32), k)
    assert got == k
    return a, b


class BlokusMCTS:
    """The reference's MCTS object for one Blokus game of the oracle: the tree is kept between searches
    (__make_root + find, mcts.hpp:385-421), clear() drops it (mcts.hpp:462-465)."""

    def __init__(self, vl=3.0, c_puct=3.0):
        L = lib()
        L.azo_blokus_mcts_new.restype = C.c_void_p
        L.azo_blokus_mcts_new.argtypes = [C.c_float, C.c_float]
        self.h = C.c_void_p(L.azo_blokus_mcts_new(vl, c_puct))

    def search(self, game, budget, noise=None):
        k = game.legal()[1]
        out = dict(actions=np.zeros(k, np.int32), n=np.zeros(k, np.int32), v=np.zeros(k, np.float32), p=np.zeros(k, np.float32))
        rn, how = C.c_int32(), C.c_int32()
        nz = None if noise is None else _p(np.ascontiguousarray(noise, np.float32), C.c_float)
        got = lib().azo_blokus_mcts_search(self.h, game.h, int(budget), nz, _p(out["actions"], C.c_int32), _p(out["n"], C.c_int32),
                                           _p(out["v"], C.c_float), _p(out["p"], C.c_float), C.byref(rn), C.byref(how))
        assert got == k, (got, k)
        out["root_n"], out["reused"] = rn.value, how.value
        return out

    def clear(self):
        lib().azo_blokus_mcts_clear(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().azo_blokus_mcts_free(self.h)
            self.h = None


def blokus_selfplay(n_moves, budget, seed, slot, vl=3.0, c_puct=3.0, reuse=False):
    """The move loop of generate.cpp:69-118 on one Blokus game of the oracle, as alphazero_amd/csrc/blokus.hip
    states it for four players: per move a search of `budget` simulations, the row (planes, to move, policy =
    n_child / n_root of the visited children in action order) buffered, the move sampled in proportion to the
    visits while the 1-based move number is below 0.2 x the running mean game length (initially 25; pick =
    (u * total) >> 32 with u = word 0 of Philox(seed ^ 'SAMP', slot, draws so far)), the first arg-max afterwards;
    a finished game releases its rows with the final reward and the next game starts.
    reuse = False: every search from a fresh root.  reuse = True: the reference's loop -- one MCTS object whose
    tree is carried from move to move (mcts.hpp:385-421), a warm-up search at every game start whose result is
    discarded (generate.cpp:77) and agent.clear() at every game end (generate.cpp:113); the warm-up takes one of
    the n_moves steps, as it does on the device where the games advance in lockstep.
    Returns (finished rows, moves played, games finished).  Test infrastructure: small cases only."""
    g = Blokus()
    agent = BlokusMCTS(vl, c_puct) if reuse else None
    avg_len, games, sample_ctr = np.float32(25.0), 0, 0
    rows, pending, moves = [], [], 0
    warm = reuse
    for _ in range(n_moves):
        if warm:
            agent.search(g, budget)                     # generate.cpp:77
            warm = False
            continue
        r = agent.search(g, budget) if reuse else blokus_search(g, budget, None, vl, c_puct)
        i = len(pending) + 1
        vis = r["n"] > 0
        pol = (r["n"][vis].astype(np.float32) / np.float32(r["root_n"])).astype(np.float32)
        pending.append(dict(planes=g.planes(), to_move=g.to_move, actions=r["actions"][vis].copy(), probs=pol))
        if float(i) < float(avg_len) * 0.2:
            total = int(r["n"].sum())
            pick = (philox(seed ^ 0x53414D50, slot, sample_ctr, 0) * total) >> 32
            sample_ctr += 1
            k = int(np.searchsorted(np.cumsum(r["n"].astype(np.int64)), pick, side="right"))
        else:
            k = int(np.argmax(r["n"]))
        rew, done = g.step(int(r["actions"][k]))
        moves += 1
        if done == 1:
            for row in pending:
                row["reward"] = rew.copy()
            rows += pending
            pending = []
            games += 1
            avg_len = np.float32(avg_len + (np.float32(i) - avg_len) / np.float32(games))
            g.reset()
            if reuse:
                agent.clear()                           # generate.cpp:113
                warm = True
    return rows, moves, games
