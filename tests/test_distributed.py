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
