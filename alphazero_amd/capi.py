"""ctypes binding of libazhip.so (C ABI: include/az_hip.h).

There is no CPU fallback: importing works anywhere (so the symbol table can be
checked without a GPU) but every compute call needs an MI355X and raises
AzError otherwise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZ_HIP_LIB") or os.path.join(_HERE, "libazhip.so")   # AZ_HIP_LIB: A/B another build

AZ_EVAL_TOWER, AZ_EVAL_HASH, AZ_EVAL_EXTERNAL = 0, 1, 2
AZ_NOISE_EXPLICIT, AZ_NOISE_DEVICE = 0, 1
AZ_PREC_FP16, AZ_PREC_BF16, AZ_PREC_FP16X3 = 0, 1, 2   # fp16 operands are the default; FP16X3 = compensated three-pass parity mode
AZ_MAX_LAYERS = 64
AZ_OK, AZ_E_ARG, AZ_E_HIP, AZ_E_CAPACITY, AZ_E_STATE, AZ_E_NOWEIGHTS = 0, -1, -2, -3, -4, -5

EXPORTS = [
    "az_last_error", "az_ctx_create", "az_ctx_destroy", "az_sync", "az_stream", "az_weights_size",
    "az_load_weights", "az_load_weights_dev", "az_weights_info_get", "az_env_reset", "az_env_step", "az_env_boards",
    "az_env_legal_mask", "az_env_set", "az_env_load", "az_tower_forward", "az_tower_forward_dev", "az_tower_forward_masked_dev", "az_set_noise",
    "az_tower_forward_static", "az_tower_static_max_batch",
    "az_group_create", "az_group_destroy", "az_group_contexts", "az_group_context", "az_group_load_weights",
    "az_group_selfplay_begin", "az_group_selfplay_run", "az_group_sync", "az_group_replay_drain", "az_group_get_stats",
    "az_replay_flip", "az_replay_drain_retired", "az_group_replay_flip", "az_group_replay_drain_retired",
    "az_ctx_cells", "az_ctx_device", "az_merge_drain", "az_device_count", "az_get_noise",
    "az_blokus_create", "az_blokus_destroy", "az_blokus_reset", "az_blokus_step", "az_blokus_legal_mask", "az_blokus_boards",
    "az_blokus_state", "az_blokus_playout", "az_blokus_sync", "az_blokus_stream", "az_blokus_orient", "az_blokus_host_state_size",
    "az_blokus_host_reset", "az_blokus_host_step", "az_blokus_host_legal", "az_blokus_host_info",
    "az_blokus_search_config", "az_blokus_search_reuse", "az_blokus_search_leaves", "az_blokus_eval_slots", "az_blokus_tree_clear", "az_blokus_selfplay_row_cap", "az_blokus_set_noise", "az_blokus_search", "az_blokus_search_begin", "az_blokus_search_round",
    "az_blokus_eval_hash", "az_blokus_eval_io", "az_blokus_search_result", "az_blokus_search_stats",
    "az_blokus_search_tower", "az_tower_shape",
    "az_blokus_noise_device", "az_blokus_selfplay_config", "az_blokus_selfplay_moves", "az_blokus_replay_drain", "az_blokus_selfplay_stats",
    "az_search", "az_search_begin", "az_search_select", "az_search_apply", "az_search_result",
    "az_tree_clear", "az_tree_dump", "az_selfplay_begin", "az_selfplay_run", "az_selfplay_run_pair", "az_pair_read", "az_selfplay_chain", "az_tree_timing", "az_tree_timing_read", "az_replay_drain",
    "az_get_stats", "az_replay_symmetries", "az_profile_enable", "az_profile_read", "az_profile_overhead", "az_tower_is_fused", "az_host_env_step", "az_selfplay_batch",
]


class AzError(RuntimeError):
    pass


class AzConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("N", C.c_int), ("K", C.c_int), ("G", C.c_int), ("n_Cs", C.c_int),
        ("Cs", C.c_int * AZ_MAX_LAYERS), ("vl", C.c_float), ("c_puct", C.c_float), ("node_cap", C.c_int),
        ("max_budget", C.c_int), ("eval_mode", C.c_int), ("noise_mode", C.c_int), ("precision", C.c_int),
        ("seed", C.c_uint64), ("game_slot_base", C.c_int), ("leaves_per_round", C.c_int),
        ("c_in", C.c_int), ("c_out", C.c_int), ("value_dim", C.c_int), ("range_fallback", C.c_int),
    ]


class AzWeightsInfo(C.Structure):
    _fields_ = [
        ("precision_in_use", C.c_int), ("fell_back", C.c_int), ("overflow_weights", C.c_int64),
        ("underflow_channels", C.c_int64), ("max_abs_weight", C.c_float), ("min_channel_max", C.c_float),
    ]


class AzStats(C.Structure):
    _fields_ = [
        ("sims", C.c_int64), ("evals", C.c_int64), ("rounds", C.c_int64), ("moves", C.c_int64),
        ("games", C.c_int64), ("replay_rows", C.c_int64), ("replay_dropped", C.c_int64),
        ("max_nodes", C.c_int32), ("max_edges", C.c_int32), ("error_games", C.c_int32), ("tree_restarts", C.c_int32),
        ("sims_backed_up", C.c_int64), ("descents", C.c_int64), ("tree_levels", C.c_int64),
        ("children_scanned", C.c_int64), ("children_expanded", C.c_int64), ("saturated", C.c_int64),
    ]


_lib = None


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps)."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    paths.add(os.path.realpath(ln.split()[-1]))
    except OSError:
        pass
    return sorted(paths)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7; libazhip.so needs the same
    soname and would otherwise pull in the system's (/opt/rocm), after which torch -- imported later -- is
    bound to a runtime it was not built with and finds no GPU.  So whichever of the two comes first, the
    runtime that gets mapped is torch's: map it here (by path, without importing torch) before libazhip."""
    import sys
    if "torch" in sys.modules or hip_runtimes_mapped():
        return                                          # a runtime is already in: the loader reuses it by soname
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return                                          # no torch here: the system runtime is the only one
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libazhip.so; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AzError("libazhip.so not built (run `make -C alphazero_amd/csrc`); there is no CPU fallback")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        rts = hip_runtimes_mapped()
        if len(rts) > 1:
            raise AzError("two HIP runtimes are mapped into this process (%s): device pointers, streams and RCCL "
                          "would not be shared between them; import alphazero_amd.capi or torch before any other "
                          "library that links libamdhip64" % ", ".join(rts))
        L.az_last_error.restype = C.c_char_p
        L.az_weights_size.restype = C.c_size_t
        L.az_stream.restype = C.c_void_p
        L.az_load_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.az_load_weights_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.az_tower_forward_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.az_tower_forward_static.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.az_tower_forward_masked_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.az_tower_static_max_batch.restype = C.c_size_t
        L.az_tower_static_max_batch.argtypes = [C.c_void_p]
        L.az_blokus_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.az_blokus_host_state_size.restype = C.c_size_t
        L.az_blokus_stream.restype = C.c_void_p
        L.az_blokus_search_config.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int]
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise AzError("libazhip error %d: %s" % (rc, load().az_last_error().decode()))
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """G lockstep games on one GPU: the reference's MCTS<Env> + Evaluator<Env>
    + PVNetwork (mcts.hpp, evaluator.hpp, network.hpp) behind one handle."""

    def __init__(self, N, K, G, Cs=None, vl=3.0, c_puct=3.0, max_budget=800, node_cap=0,
                 eval_mode=AZ_EVAL_TOWER, noise_mode=AZ_NOISE_EXPLICIT, precision=AZ_PREC_FP16,
                 seed=0, device=0, game_slot_base=0, leaves_per_round=1, c_in=0, c_out=0, value_dim=0,
                 range_fallback=0):
        L = load()
        cfg = AzConfig()
        cfg.device, cfg.N, cfg.K, cfg.G = device, N, K, G
        Cs = list(Cs) if Cs is not None else []
        cfg.n_Cs = len(Cs)
        for i, c in enumerate(Cs):
            cfg.Cs[i] = c
        cfg.vl, cfg.c_puct = vl, c_puct
        cfg.node_cap, cfg.max_budget = node_cap, max_budget
        cfg.eval_mode, cfg.noise_mode, cfg.precision = eval_mode, noise_mode, precision
        cfg.seed, cfg.game_slot_base = seed, game_slot_base
        cfg.leaves_per_round = leaves_per_round
        cfg.c_in, cfg.c_out, cfg.value_dim = c_in, c_out, value_dim
        cfg.range_fallback = range_fallback
        self.c_in, self.c_out, self.value_dim = c_in or 3, c_out or 1, value_dim or 2
        self.L = max(1, leaves_per_round)
        self.N, self.K, self.G, self.A = N, K, G, N * N
        self.Cs = Cs
        self.h = C.c_void_p()
        _chk(L.az_ctx_create(C.byref(self.h), C.byref(cfg)))

    def close(self):
        if getattr(self, "h", None):
            load().az_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    # -- weights (util.h load_network) -------------------------------------
    def weights_size(self):
        return load().az_weights_size(self.h)

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, np.float32)
        _chk(load().az_load_weights(self.h, _ptr(blob), blob.size))

    def load_weights_dev(self, dev_ptr, n_floats):
        _chk(load().az_load_weights_dev(self.h, C.c_void_p(dev_ptr), n_floats))

    def weights_info(self):
        """Where the last blob's folded weights sit in fp16's range and the operand format in use."""
        w = AzWeightsInfo()
        _chk(load().az_weights_info_get(self.h, C.byref(w)))
        return {k: getattr(w, k) for k, _ in AzWeightsInfo._fields_}

    # -- PVNetwork::forward ------------------------------------------------
    def tower_forward(self, boards):
        boards = np.ascontiguousarray(boards, np.uint8)
        B = boards.shape[0]
        pol = np.empty((B, self.c_out * self.A), np.float32)
        val = np.empty((B, self.value_dim), np.float32)
        _chk(load().az_tower_forward(self.h, _ptr(boards), B, _ptr(pol), _ptr(val)))
        return pol, val

    def tower_forward_dev(self, boards_ptr, B, policy_ptr, value_ptr):
        _chk(load().az_tower_forward_dev(self.h, C.c_void_p(boards_ptr), B, C.c_void_p(policy_ptr), C.c_void_p(value_ptr)))

    def tower_forward_masked_dev(self, boards_ptr, B, policy_ptr, value_ptr, wanted_ptr):
        """only the policy entries whose bit is set in wanted u32 [B, c_out * N] are written"""
        _chk(load().az_tower_forward_masked_dev(self.h, C.c_void_p(boards_ptr), B, C.c_void_p(policy_ptr), C.c_void_p(value_ptr),
                                                C.c_void_p(wanted_ptr)))

    def tower_forward_static(self, boards_ptr, B, policy_ptr, value_ptr):
        """az_tower_forward_dev without any allocation: activations in the library's per-GPU static array."""
        _chk(load().az_tower_forward_static(self.h, C.c_void_p(boards_ptr), B, C.c_void_p(policy_ptr), C.c_void_p(value_ptr)))

    def tower_static_max_batch(self):
        return int(load().az_tower_static_max_batch(self.h))

    # -- Env ---------------------------------------------------------------
    def env_reset(self):
        _chk(load().az_env_reset(self.h))

    def env_step(self, actions):
        actions = np.ascontiguousarray(actions, np.int32)
        reward = np.empty((self.G, 2), np.float32)
        done = np.empty(self.G, np.uint8)
        _chk(load().az_env_step(self.h, _ptr(actions), _ptr(reward), _ptr(done)))
        return reward, done

    def env_boards(self):
        pl = np.empty((self.G, 3, self.N, self.N), np.uint8)
        _chk(load().az_env_boards(self.h, _ptr(pl)))
        return pl

    def env_legal_mask(self):
        m = np.empty((self.G, self.N, self.N), np.uint8)
        _chk(load().az_env_legal_mask(self.h, _ptr(m)))
        return m

    def env_set(self, cells, to_move):
        cells = np.ascontiguousarray(cells, np.uint8)
        to_move = np.ascontiguousarray(to_move, np.uint8)
        _chk(load().az_env_set(self.h, _ptr(cells), _ptr(to_move)))

    def env_load(self, histories):
        """histories: list of G move lists (cell indices)."""
        hist = np.zeros((self.G, self.A), np.int32)
        hl = np.zeros(self.G, np.int32)
        for g, h in enumerate(histories):
            hl[g] = len(h)
            hist[g, :len(h)] = h
        _chk(load().az_env_load(self.h, _ptr(hist), _ptr(hl)))

    # -- MCTS --------------------------------------------------------------
    def set_noise(self, noise):
        noise = np.ascontiguousarray(noise, np.float32)
        assert noise.size == self.G * self.A
        _chk(load().az_set_noise(self.h, _ptr(noise)))

    def get_noise(self):
        out = np.empty((self.G, self.A), np.float32)
        _chk(load().az_get_noise(self.h, _ptr(out)))
        return out

    def search(self, budget):
        out = np.empty((self.G, self.A), np.float32)
        _chk(load().az_search(self.h, int(budget), _ptr(out)))
        return out

    def search_begin(self, budget):
        _chk(load().az_search_begin(self.h, int(budget)))

    def search_select(self):
        planes = np.empty((self.G * self.L, 3, self.N, self.N), np.uint8)
        pending = np.empty(self.G * self.L, np.int32)
        run = C.c_int()
        _chk(load().az_search_select(self.h, _ptr(planes), _ptr(pending), C.byref(run)))
        return planes, pending, run.value

    def search_apply(self, policy, value):
        policy = np.ascontiguousarray(policy, np.float32)
        value = np.ascontiguousarray(value, np.float32)
        assert policy.size == self.G * self.L * self.A and value.size == self.G * self.L * 2
        _chk(load().az_search_apply(self.h, _ptr(policy), _ptr(value)))

    def search_result(self):
        out = np.empty((self.G, self.A), np.float32)
        rn = np.empty(self.G, np.int32)
        _chk(load().az_search_result(self.h, _ptr(out), _ptr(rn)))
        return out, rn

    def tree_clear(self):
        _chk(load().az_tree_clear(self.h))

    def tree_dump(self, game):
        L = load()
        nn, ne = C.c_int32(), C.c_int32()
        _chk(L.az_tree_dump(self.h, game, 0, 0, C.byref(nn), C.byref(ne), None, None, None, None, None, None, None))
        nn, ne = nn.value, ne.value
        d = dict(node_first=np.zeros(nn, np.int32), node_count=np.zeros(nn, np.int32), p=np.zeros(ne, np.float32),
                 n=np.zeros(ne, np.int32), v=np.zeros(ne, np.float32), child=np.zeros(ne, np.int32),
                 action=np.zeros(ne, np.int32))
        a, b = C.c_int32(), C.c_int32()
        if nn:
            _chk(L.az_tree_dump(self.h, game, nn, ne, C.byref(a), C.byref(b), _ptr(d["node_first"]), _ptr(d["node_count"]),
                                _ptr(d["p"]), _ptr(d["n"]), _ptr(d["v"]), _ptr(d["child"]), _ptr(d["action"])))
        return d

    def tree_canonical(self, game):
        """BFS dump in the oracle's format: (parent, action, n, v, p, kind)."""
        d = self.tree_dump(game)
        out = []
        if len(d["node_first"]) == 0:
            return out
        q = [0]
        qi = 0
        while qi < len(q):
            node = q[qi]
            fe, ne = int(d["node_first"][node]), int(d["node_count"][node])
            for e in range(fe, fe + ne):
                ch = int(d["child"][e])
                kind = 1 if ch >= 0 else {-1: 0, -2: 2, -3: 3, -4: 4}[ch]
                out.append((qi, int(d["action"][e]), int(d["n"][e]), float(d["v"][e]), float(d["p"][e]), kind))
                if ch >= 0:
                    q.append(ch)
            qi += 1
        return out

    # -- generate.cpp on the device ------------------------------------------
    def selfplay_begin(self, budget):
        _chk(load().az_selfplay_begin(self.h, int(budget)))

    def selfplay_run(self, rounds):
        _chk(load().az_selfplay_run(self.h, int(rounds)))

    def selfplay_run_pair(self, rounds):
        """The same rounds as one launch of the persistent tower / tree pair."""
        _chk(load().az_selfplay_run_pair(self.h, int(rounds)))

    def selfplay_chain(self, max_chained):
        """simulations that end in a stored terminal leaf (no evaluation) a round may run back to back"""
        _chk(load().az_selfplay_chain(self.h, int(max_chained)))

    def tree_timing(self, on):
        _chk(load().az_tree_timing(self.h, int(bool(on))))

    def tree_timing_read(self):
        v = [C.c_double() for _ in range(5)]
        n = C.c_int64()
        _chk(load().az_tree_timing_read(self.h, *[C.byref(x) for x in v], C.byref(n)))
        return dict(zip(("select_us", "step_us", "append_us", "backup_us", "root_us"), [x.value for x in v]), steps=n.value)

    def pair_read(self):
        us, n, r, ts = C.c_double(), C.c_int64(), C.c_int64(), C.c_double()
        _chk(load().az_pair_read(self.h, C.byref(us), C.byref(n), C.byref(r), C.byref(ts)))
        return {"launch_avg_us": us.value, "launches": n.value, "rounds": r.value, "tree_step_avg_us": ts.value}

    def replay_drain(self, cap=1 << 16):
        cells = np.empty((cap, self.A), np.uint8)
        tm = np.empty(cap, np.uint8)
        pol = np.empty((cap, self.A), np.float32)
        rew = np.empty((cap, 2), np.float32)
        n = _chk(load().az_replay_drain(self.h, cap, _ptr(cells), _ptr(tm), _ptr(pol), _ptr(rew)))
        return cells[:n], tm[:n], pol[:n], rew[:n]

    def replay_flip(self):
        """retire the ring the launches queued so far append to; later launches append to the other one"""
        _chk(load().az_replay_flip(self.h))

    def replay_drain_retired(self, cap=1 << 16):
        """rows of the retired ring, waiting only for the launches queued before the flip"""
        cells = np.empty((cap, self.A), np.uint8)
        tm = np.empty(cap, np.uint8)
        pol = np.empty((cap, self.A), np.float32)
        rew = np.empty((cap, 2), np.float32)
        n = _chk(load().az_replay_drain_retired(self.h, cap, _ptr(cells), _ptr(tm), _ptr(pol), _ptr(rew)))
        return cells[:n], tm[:n], pol[:n], rew[:n]

    def stats(self):
        s = AzStats()
        _chk(load().az_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in AzStats._fields_}

    def sync(self):
        _chk(load().az_sync(self.h))

    def tower_is_fused(self):
        return bool(load().az_tower_is_fused(self.h))

    def selfplay_batch(self):
        return int(load().az_selfplay_batch(self.h))

    def profile_enable(self, every):
        _chk(load().az_profile_enable(self.h, int(every)))

    def profile_read(self):
        ca, ta = C.c_double(), C.c_double()
        cn, tn = C.c_int64(), C.c_int64()
        _chk(load().az_profile_read(self.h, C.byref(ca), C.byref(cn), C.byref(ta), C.byref(tn)))
        ov, on = C.c_double(), C.c_int64()
        _chk(load().az_profile_overhead(self.h, C.byref(ov), C.byref(on)))
        return dict(conv_avg_us=ca.value, conv_launches=cn.value, tree_avg_us=ta.value, tree_launches=tn.value,
                    empty_span_us=ov.value, empty_spans=on.value)


def host_env_step(N, K, cells, to_move, action):
    cells = np.ascontiguousarray(cells, np.uint8).copy()
    tm = C.c_int(int(to_move))
    rew = np.zeros(2, np.float32)
    rc = load().az_host_env_step(N, K, _ptr(cells), C.byref(tm), int(action), _ptr(rew))
    return cells, tm.value, rew, rc


def replay_symmetries(N, cells, to_move, policy):
    cells = np.ascontiguousarray(cells, np.uint8)
    policy = np.ascontiguousarray(policy, np.float32)
    so = np.empty((8, 3, N, N), np.uint8)
    po = np.empty((8, N, N), np.float32)
    _chk(load().az_replay_symmetries(N, _ptr(cells), int(to_move), _ptr(policy), _ptr(so), _ptr(po)))
    return so, po


class Blokus:
    """G lockstep Blokus games on one GPU (az_blokus_*): 20x20, four players, 91 orientations,
    action = o * 400 + y * 20 + x, planes u8 [5,20,20], reward f32 [4]."""
    N, ORIENTS, CELLS, ACTIONS, PLAYERS, PLANES = 20, 91, 400, 91 * 400, 4, 5

    def __init__(self, G, device=0, seed=0, game_slot_base=0):
        self.G = G
        self.h = C.c_void_p()
        _chk(load().az_blokus_create(C.byref(self.h), device, G, seed, game_slot_base))

    def close(self):
        if getattr(self, "h", None):
            load().az_blokus_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        _chk(load().az_blokus_reset(self.h))

    def step(self, actions):
        actions = np.ascontiguousarray(actions, np.int32)
        reward = np.empty((self.G, 4), np.float32)
        done = np.empty(self.G, np.uint8)
        _chk(load().az_blokus_step(self.h, _ptr(actions), _ptr(reward), _ptr(done)))
        return reward, done

    def legal_mask(self, want_mask=True):
        mask = np.empty((self.G, self.ACTIONS), np.uint8) if want_mask else None
        count = np.empty(self.G, np.int32)
        _chk(load().az_blokus_legal_mask(self.h, _ptr(mask) if want_mask else None, _ptr(count)))
        return mask, count

    def boards(self):
        pl = np.empty((self.G, self.PLANES, self.N, self.N), np.uint8)
        _chk(load().az_blokus_boards(self.h, _ptr(pl)))
        return pl

    def state(self):
        tm, over = np.empty(self.G, np.uint8), np.empty(self.G, np.uint8)
        moves = np.empty(self.G, np.int32)
        used = np.empty((self.G, 4), np.uint32)
        fr = np.empty((self.G, 4), np.float32)
        _chk(load().az_blokus_state(self.h, _ptr(tm), _ptr(over), _ptr(moves), _ptr(used), _ptr(fr)))
        return dict(to_move=tm, over=over, moves=moves, used=used, final_reward=fr)

    def playout(self, steps):
        _chk(load().az_blokus_playout(self.h, int(steps)))

    # -- MCTS ----------------------------------------------------------------
    def search_config(self, vl=3.0, c_puct=3.0, max_budget=400, edge_cap=0, reuse=False, leaves=1):
        """reuse: keep each game's tree between searches (__make_root + find, mcts.hpp:385-421); self-play then runs
        the reference's loop in full (warm-up search per game, tree carried from move to move).  leaves: descents per
        game and round (MCTS(nthreads, ...), mcts.hpp:54-72): a round evaluates G * leaves positions"""
        if reuse:
            _chk(load().az_blokus_search_reuse(self.h, 1))
        if leaves != 1:
            _chk(load().az_blokus_search_leaves(self.h, int(leaves)))
        self.leaves = int(leaves)
        _chk(load().az_blokus_search_config(self.h, vl, c_puct, int(max_budget), int(edge_cap)))
        self.K = int(max_budget)                    # policy entries kept per self-play row (selfplay_config refreshes it)

    def tree_clear(self):
        _chk(load().az_blokus_tree_clear(self.h))

    def set_noise(self, noise):
        if noise is None:
            _chk(load().az_blokus_set_noise(self.h, None, 0))
            return
        noise = np.ascontiguousarray(noise, np.float32)
        assert noise.shape[0] == self.G
        _chk(load().az_blokus_set_noise(self.h, _ptr(noise), noise.shape[1]))

    def search(self, budget):
        _chk(load().az_blokus_search(self.h, int(budget)))

    def search_tower(self, tower_engine, budget):
        """search with the leaves evaluated by `tower_engine` (an Engine with N=20, c_in=5, c_out=91, value_dim=4)"""
        _chk(load().az_blokus_search_tower(self.h, tower_engine.h, int(budget)))

    def eval_io(self):
        pl, po, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(load().az_blokus_eval_io(self.h, C.byref(pl), C.byref(po), C.byref(va)))
        return pl.value, po.value, va.value

    def search_result(self, game):
        L = load()
        nc, rn = C.c_int32(), C.c_int32()
        _chk(L.az_blokus_search_result(self.h, game, 0, C.byref(nc), C.byref(rn), None, None, None, None))
        k = nc.value
        out = dict(actions=np.zeros(k, np.int32), n=np.zeros(k, np.int32), v=np.zeros(k, np.float32), p=np.zeros(k, np.float32))
        if k:
            _chk(L.az_blokus_search_result(self.h, game, k, C.byref(nc), C.byref(rn), _ptr(out["actions"]), _ptr(out["n"]),
                                           _ptr(out["v"]), _ptr(out["p"])))
        out["root_n"] = rn.value
        return out

    def search_stats(self):
        mn, ms, rs = C.c_int32(), C.c_int32(), C.c_int32()
        rd = C.c_int64()
        _chk(load().az_blokus_search_stats(self.h, C.byref(mn), C.byref(ms), C.byref(rs), C.byref(rd)))
        return dict(max_nodes=mn.value, max_slots=ms.value, restarts=rs.value, rounds=rd.value)

    def noise_device(self, on=True):
        _chk(load().az_blokus_noise_device(self.h, int(bool(on))))

    # -- self-play -------------------------------------------------------------
    def selfplay_config(self, ring_rows, max_budget=None):
        """after search_config; max_budget, if given, must be the value search_config got"""
        assert max_budget is None or int(max_budget) == getattr(self, "K", None)
        _chk(load().az_blokus_selfplay_config(self.h, int(ring_rows)))
        self.K = int(load().az_blokus_selfplay_row_cap(self.h))

    def selfplay_moves(self, budget, n_moves, tower_engine=None):
        _chk(load().az_blokus_selfplay_moves(self.h, tower_engine.h if tower_engine is not None else None, int(budget), int(n_moves)))

    def replay_drain(self, cap):
        """finished rows, oldest first: planes u8 [n,5,20,20], to_move u8 [n], counts i32 [n], actions i32 / probs f32
        [n,K] (the visited root children in action order; -1 / 0 beyond counts), reward f32 [n,4]"""
        K = self.K
        pl = np.empty((cap, self.PLANES, self.N, self.N), np.uint8)
        tm = np.empty(cap, np.uint8)
        cnt = np.empty(cap, np.int32)
        act = np.empty((cap, K), np.int32)
        pr = np.empty((cap, K), np.float32)
        rw = np.empty((cap, 4), np.float32)
        n = C.c_int32()
        _chk(load().az_blokus_replay_drain(self.h, int(cap), _ptr(pl), _ptr(tm), _ptr(cnt), _ptr(act), _ptr(pr), _ptr(rw), C.byref(n)))
        n = n.value
        return dict(planes=pl[:n], to_move=tm[:n], counts=cnt[:n], actions=act[:n], probs=pr[:n], reward=rw[:n])

    @staticmethod
    def dense_policy(actions, probs, counts):
        """the rows' policies in the Env's action shape [n, 91, 20, 20]"""
        out = np.zeros((len(counts), Blokus.ACTIONS), np.float32)
        for r, c in enumerate(counts):
            out[r, actions[r, :c]] = probs[r, :c]
        return out.reshape(-1, Blokus.ORIENTS, Blokus.N, Blokus.N)

    def selfplay_stats(self):
        mv, gm, rw, dr = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _chk(load().az_blokus_selfplay_stats(self.h, C.byref(mv), C.byref(gm), C.byref(rw), C.byref(dr)))
        return dict(moves=mv.value, games=gm.value, rows=rw.value, dropped=dr.value)

    def sync(self):
        _chk(load().az_blokus_sync(self.h))


def blokus_orient(o):
    piece, h, w = C.c_int(), C.c_int(), C.c_int()
    cells = np.zeros((5, 2), np.int8)
    n = _chk(load().az_blokus_orient(int(o), C.byref(piece), C.byref(h), C.byref(w), _ptr(cells)))
    return piece.value, h.value, w.value, [tuple(int(v) for v in c) for c in cells[:n]]


class BlokusHost:
    """One Blokus game on the host (the library's CPU twin of its device rules; no GPU needed)."""

    def __init__(self):
        self.buf = C.create_string_buffer(load().az_blokus_host_state_size())
        _chk(load().az_blokus_host_reset(self.buf))

    def step(self, action):
        rew = np.zeros(4, np.float32)
        done = _chk(load().az_blokus_host_step(self.buf, int(action), _ptr(rew)))
        return rew, done

    def legal(self):
        mask = np.zeros(91 * 400, np.uint8)
        n = _chk(load().az_blokus_host_legal(self.buf, _ptr(mask)))
        return mask, n

    def info(self):
        cells = np.zeros(400, np.uint8)
        tm, over = C.c_int(), C.c_int()
        used = np.zeros(4, np.uint32)
        scores = np.zeros(4, np.int32)
        _chk(load().az_blokus_host_info(self.buf, _ptr(cells), C.byref(tm), C.byref(over), _ptr(used), _ptr(scores)))
        return dict(cells=cells, to_move=tm.value, over=bool(over.value), used=used, scores=scores)
