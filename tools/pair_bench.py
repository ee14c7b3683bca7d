#!/usr/bin/env python3
"""Same-device A/B of the self-play rounds as lockstep launches (az_selfplay_run) and as the persistent tower /
tree pair (az_selfplay_run_pair), legs alternating in one process.

    python tools/pair_bench.py [--rounds 6400] [--modes lockstep,pair,pair2,pair5]

A mode `pairF` runs the pair with AZ_PAIR_FLAGS=F (timing experiments, see az_selfplay_run_pair in api.hip):
2 = no early look at the next tile's hand-over words, 5 = the tower alone (no tree kernel, never waits: wrong
games, the time a round would take if the tree cost nothing)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--budget", type=int, default=800)
    ap.add_argument("--rounds", type=int, default=6400)
    ap.add_argument("--modes", default="lockstep,pair,lockstep,pair,pair5")
    a = ap.parse_args()
    Cs = synth.NETCONFIG2
    blob = synth.make_blob(a.board, 3, Cs, 0)
    for mode in a.modes.split(","):
        e = capi.Engine(a.board, 5, a.games, Cs=Cs, max_budget=a.budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=0)
        e.load_weights(blob)
        e.selfplay_begin(a.budget)
        os.environ["AZ_PAIR_FLAGS"] = mode[4:] if mode.startswith("pair") and len(mode) > 4 else "0"
        run = e.selfplay_run if mode == "lockstep" else e.selfplay_run_pair
        run(a.budget)
        e.profile_enable(1 if mode.startswith("pair") else 9)
        e.sync()
        s0 = e.stats()["sims_backed_up"]
        t0 = time.perf_counter()
        for _ in range(a.rounds // a.budget):
            run(a.budget)
        e.sync()
        el = time.perf_counter() - t0
        line = {"mode": mode, "sims_per_s": (e.stats()["sims_backed_up"] - s0) / el, "us_per_round": el / a.rounds * 1e6}
        line.update(e.pair_read() if mode.startswith("pair") else e.profile_read())
        print(line, flush=True)
        e.close()


if __name__ == "__main__":
    main()
