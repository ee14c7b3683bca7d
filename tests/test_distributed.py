"""world_size-2 gloo tests of the multi-GPU plumbing (weight broadcast, replay
row gather, slot sharding) on CPU; the data path itself has no collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alphazero_amd import selfplay, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # slot sharding: disjoint, contiguous, covering
        lo, hi = selfplay.shard_slots(rank, world, 512)
        assert (lo, hi) == (rank * 512, rank * 512 + 512)
        # weight broadcast: every rank ends up with rank 0's blob, bit for bit
        N, Cs = 5, [16, 16, 16]
        n = synth.blob_size(N, 3, Cs)
        blob = synth.make_blob(N, 3, Cs, 3) if rank == 0 else None
        t = selfplay.broadcast_weights(blob, n, 0)
        assert np.array_equal(t.numpy(), synth.make_blob(N, 3, Cs, 3))
        # ragged row gather (rank r contributes 3*r+1 rows; rank order preserved)
        A = N * N
        k = 3 * rank + 1
        rng = np.random.default_rng(rank)
        cells = rng.integers(0, 3, (k, A)).astype(np.uint8)
        tm = rng.integers(0, 2, k).astype(np.uint8)
        pol = rng.random((k, A)).astype(np.float32)
        rew = rng.random((k, 2)).astype(np.float32)
        got = selfplay.gather_rows(cells, tm, pol, rew, dst=0)
        if rank == 0:
            want = [[], [], [], []]
            for r in range(world):
                g = np.random.default_rng(r)
                kk = 3 * r + 1
                want[0].append(g.integers(0, 3, (kk, A)).astype(np.uint8))
                want[1].append(g.integers(0, 2, kk).astype(np.uint8))
                want[2].append(g.random((kk, A)).astype(np.float32))
                want[3].append(g.random((kk, 2)).astype(np.float32))
            for a, w in zip(got, want):
                assert np.array_equal(a, np.concatenate(w, 0))
        else:
            assert got is None
        # empty contribution from one rank
        e = selfplay.gather_rows(cells[:0] if rank == 1 else cells, tm[:0] if rank == 1 else tm,
                                 pol[:0] if rank == 1 else pol, rew[:0] if rank == 1 else rew, dst=0)
        if rank == 0:
            assert len(e[1]) == 1
        q.put((rank, "ok"))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: "ok", 1: "ok"}, res


def test_merge_drain_orders_rows_by_source_with_two_fake_ranks():
    """az_merge_drain (the host-side merge az_group_replay_drain is made of) with two fake GPU ranks as
    callbacks: rows come out source-major and in each source's own order, a cap smaller than the total
    leaves the rest in its source for the next call, a source's error ends the merge."""
    import ctypes as C
    from alphazero_amd import capi
    L = capi.load()
    A = 9
    FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                     C.POINTER(C.c_float), C.POINTER(C.c_float))
    L.az_merge_drain.argtypes = [C.c_int, FN, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    # fake rank r holds rows tagged (r, i): cells all r + 1, to_move i % 2, policy i / 100, reward (r, -r)
    pending = {0: list(range(5)), 1: list(range(7))}
    calls = []

    def drain(src, cap, cells, tm, pol, rew):
        r = int(src or 0) - 1
        if r == 2:
            return -4
        calls.append((r, cap))
        n = min(cap, len(pending[r]))
        for k in range(n):
            i = pending[r].pop(0)
            for a in range(A):
                cells[k * A + a] = r + 1
                pol[k * A + a] = i / 100.0
            tm[k] = i % 2
            rew[2 * k], rew[2 * k + 1] = float(r), float(-r)
        return n
    fn = FN(drain)
    srcs = (C.c_void_p * 2)(1, 2)                              # opaque handles: rank + 1

    def merge(cap, srcs=srcs, n=2):
        cells = np.zeros((cap, A), np.uint8); tm = np.zeros(cap, np.uint8)
        pol = np.zeros((cap, A), np.float32); rew = np.zeros((cap, 2), np.float32)
        k = L.az_merge_drain(n, fn, srcs, A, cap, cells.ctypes.data, tm.ctypes.data, pol.ctypes.data, rew.ctypes.data)
        return k, cells[:max(k, 0)], tm[:max(k, 0)], pol[:max(k, 0)], rew[:max(k, 0)]
    k, cells, tm, pol, rew = merge(8)                          # 5 rows of rank 0, then 3 of rank 1
    assert k == 8 and calls == [(0, 8), (1, 3)]
    assert cells[:, 0].tolist() == [1] * 5 + [2] * 3
    assert np.allclose(pol[:, 0], [0, .01, .02, .03, .04, 0, .01, .02])
    assert rew[:, 0].tolist() == [0] * 5 + [1] * 3 and tm.tolist() == [0, 1, 0, 1, 0, 0, 1, 0]
    k, cells, tm, pol, rew = merge(100)                        # the rest: rank 0 is empty, rank 1 has 4 left
    assert k == 4 and cells[:, 0].tolist() == [2] * 4 and np.allclose(pol[:, 0], [.03, .04, .05, .06])
    assert merge(100)[0] == 0
    assert merge(0)[0] == 0
    bad = (C.c_void_p * 2)(1, 3)
    pending[0] = [0]
    assert merge(10, bad)[0] == -4                             # the failing source's code comes back
