"""Multi-GPU self-play driver: one process per GPU, games sharded by rank.

The hot path shards by game (trees never interact), so there is no collective
in the data path.  `torch.distributed` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests) is used for exactly two hand-offs:
  * broadcast_weights: the flat fp32 weight blob from the rank that read the
    new checkpoint (generate.cpp:73 / util.h:60-105) to every rank;
  * gather_rows: finished replay rows (replay.hpp:56-60) to the rank that owns
    the PUSH socket (replay.hpp:41-54).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_slots(rank, world, games_per_gpu):
    """Global slot ids of rank's games: [rank*G, (rank+1)*G)."""
    return rank * games_per_gpu, (rank + 1) * games_per_gpu


def broadcast_weights(blob, n_floats, src=0, device="cpu"):
    """blob: np.float32 array on rank src, None elsewhere.  Returns a tensor on `device`."""
    t = torch.empty(n_floats, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(blob, np.float32)))
    if dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return t


def gather_rows(cells, to_move, policy, reward, dst=0, device="cpu"):
    """Variable-length gather of replay rows to rank dst.  Each argument is this
    rank's np array with a leading row dimension.  Returns the concatenated
    arrays on dst (rank order, row order preserved) and None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(to_move)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    A = cells.shape[1]
    # one fixed-size slab per rank: [cells u8 A | to_move u8 1 | policy f32 A | reward f32 2] as bytes
    row_bytes = A + 1 + 4 * A + 8
    slab = np.zeros((mx, row_bytes), np.uint8)
    k = len(to_move)
    if k:
        slab[:k, :A] = cells
        slab[:k, A] = to_move
        slab[:k, A + 1:A + 1 + 4 * A] = np.ascontiguousarray(policy, np.float32).view(np.uint8).reshape(k, 4 * A)
        slab[:k, A + 1 + 4 * A:] = np.ascontiguousarray(reward, np.float32).view(np.uint8).reshape(k, 8)
    mine = torch.from_numpy(slab).to(device)
    if rank == dst:
        bufs = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, bufs, dst=dst)
        parts = [b.cpu().numpy()[:c] for b, c in zip(bufs, counts)]
        allr = np.concatenate(parts, 0) if parts else slab[:0]
        out_cells = allr[:, :A].copy()
        out_tm = allr[:, A].copy()
        out_pol = allr[:, A + 1:A + 1 + 4 * A].copy().view(np.float32).reshape(-1, A)
        out_rew = allr[:, A + 1 + 4 * A:].copy().view(np.float32).reshape(-1, 2)
        return out_cells, out_tm, out_pol, out_rew
    dist.gather(mine, None, dst=dst)
    return None


def run_selfplay(N, K, games_per_gpu, budget, Cs, blob_on_rank0, rounds, local_rank=0, seed=0, sink=None, contexts=1):
    """Rank-sharded device self-play.  blob_on_rank0: weights on rank 0 (None
    elsewhere).  sink(cells, to_move, policy, reward) is called on rank 0 with
    the gathered rows.  Needs an MI355X per rank.

    A rank's games run as `contexts` engine contexts on their own HIP streams.  With
    AZ_PAIR=1 in the environment (one context, the one-launch tower) the rounds of a
    move run as one launch of the persistent tower / tree pair (az_selfplay_run_pair)."""
    from . import capi
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, _ = shard_slots(rank, world, games_per_gpu)
    if contexts < 1 or games_per_gpu < 2 * contexts:
        contexts = 1
    sizes = [games_per_gpu // contexts + (1 if i < games_per_gpu % contexts else 0) for i in range(contexts)]
    engs, first = [], lo
    for n in sizes:
        engs.append(capi.Engine(N, K, n, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=seed,
                                device=local_rank, game_slot_base=first))
        first += n
    wt = broadcast_weights(blob_on_rank0, engs[0].weights_size(), 0, torch.device("cuda", local_rank))
    torch.cuda.synchronize()
    for e in engs:
        e.load_weights_dev(wt.data_ptr(), wt.numel())
        e.selfplay_begin(budget)
    import os
    slice_rounds = max(1, budget) if os.environ.get("AZ_PAIR", "0") not in ("", "0") else 16
    done = 0
    while done < rounds:                                   # slices keep every stream fed (a pair launch: a move's rounds)
        k = min(slice_rounds, rounds - done)
        for e in engs:
            e.selfplay_run(k)
        done += k
    parts = [e.replay_drain() for e in engs]
    rows = tuple(np.concatenate([p[i] for p in parts], 0) for i in range(4))
    got = gather_rows(*rows, dst=0, device=torch.device("cuda", local_rank))
    if got is not None and sink is not None:
        sink(*got)
    sts = [e.stats() for e in engs]
    st = {k: (max if k in ("max_nodes", "max_edges", "rounds") else sum)(x[k] for x in sts) for k in sts[0]}
    for e in engs:
        e.close()
    return st
