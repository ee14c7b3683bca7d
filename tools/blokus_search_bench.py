#!/usr/bin/env python3
"""BASELINE configs[3]: Blokus 20x20 four players, 2048 concurrent games x 400 simulations per search, on the
device's tree kernel with the integer-hash evaluator (no Blokus tower exists yet: the number is the env + tree
path alone).  One JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel summary."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=2048)
    ap.add_argument("--budget", type=int, default=400)
    ap.add_argument("--pieces", type=int, default=9, help="random pieces on the board before the search")
    ap.add_argument("--searches", type=int, default=3)
    a = ap.parse_args()
    from alphazero_amd import capi
    dev = capi.Blokus(a.games, seed=3)
    dev.search_config(max_budget=a.budget)
    dev.playout(a.pieces)
    dev.sync()
    dev.search(8)                                   # warm-up
    t0 = time.perf_counter()
    for _ in range(a.searches):
        dev.search(a.budget)
    dev.sync()
    el = time.perf_counter() - t0
    st = dev.search_stats()
    _, count = dev.legal_mask(want_mask=False)
    sims = a.games * a.budget * a.searches
    print(json.dumps({"metric": "mcts_simulations_per_sec", "value": sims / el, "unit": "sims/s",
                      "config": {"workload": "blokus 20x20 4-player, %d concurrent games x %d sims per search, integer-hash evaluator "
                                             "(env + tree kernels only; no Blokus tower)" % (a.games, a.budget),
                                 "pieces_on_board": a.pieces, "mean_legal_actions_at_root": float(count.mean())},
                      "ms_per_round": el / (a.searches * (a.budget + 3)) * 1e3, "searches": a.searches, "seconds": el, "tree": st}))


if __name__ == "__main__":
    main()
