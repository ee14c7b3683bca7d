#!/usr/bin/env python3
"""Env-only measurement of the Blokus kernels (BASELINE configs[3] size: 2048 concurrent games): whole games
of uniformly random legal moves on the device, one launch per move of every game.  Prints one JSON line with
moves/s, the launch duration from HIP events on the library's stream, and the achieved rate in ALGORITHMIC
bytes: per step a game reads and writes the reference-style u8 [5,20,20] board (2 x 2 000 B) and produces the
legal mask of the next player as bits (91 x 400 / 8 = 4 550 B) plus 16 B of bookkeeping = 8 566 B.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel summary kept in profiles/."""
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
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    import torch
    from alphazero_amd import capi
    G = a.games
    dev = capi.Blokus(G, seed=1)
    stream = torch.cuda.ExternalStream(capi.load().az_blokus_stream(dev.h))
    total_moves, total_s, launch_us = 0, 0.0, []
    for rep in range(a.repeats):
        dev.reset()
        t0 = time.perf_counter()
        for _ in range(84):                       # no game is longer than 84 pieces
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dev.playout(1)
            e1.record(stream)
            launch_us.append((e0, e1))
        dev.sync()
        total_s += time.perf_counter() - t0
        total_moves += int(dev.state()["moves"].sum())
    first = [e0.elapsed_time(e1) * 1e3 for e0, e1 in launch_us[:40]] + [e0.elapsed_time(e1) * 1e3 for e0, e1 in launch_us[84:124]]
    us = float(sum(first) / len(first))           # launches in which (almost) every game still moves
    bytes_per_step = 2 * 5 * 400 + 91 * 400 // 8 + 16
    gbs = G * bytes_per_step / (us * 1e-6) / 1e9
    print(json.dumps({"metric": "blokus_env_moves_per_sec", "value": total_moves / total_s, "unit": "moves/s",
                      "config": {"workload": "blokus 20x20 4-player, %d concurrent games, uniformly random legal moves" % G},
                      "avg_launch_us_first_40_moves": us, "algorithmic_bytes_per_step": bytes_per_step,
                      "roofline": {"bound": "hbm", "kernel": "azb_playout_kernel", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                   "frac": gbs / 8000.0},
                      "moves": total_moves, "seconds": total_s}))


if __name__ == "__main__":
    main()
