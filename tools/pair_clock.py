#!/usr/bin/env python3
"""Shader clock of the fused tower under lockstep launches (tower, tree, tower, ...) and as the persistent kernel
with no gaps at all: cycles and wall time of one tile from the stamps of an -DAZ_TOWER_TIMING build
(alphazero_amd/libazhip_timing.so; build: tower.hip with -DAZ_TOWER_TIMING -DAZ_TOWER_REG_CAP=).
AZ_PAIR_FLAGS=5 must be set: the timing build is too large to share a CU with a tree wave, so the persistent
leg runs the tower alone on unchanging planes."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "alphazero_amd", "libazhip_timing.so")


def stamps():
    ts = np.zeros(128, np.uint64)
    assert capi.load().az_tower_timing_read(ts.ctypes.data_as(C.c_void_p)) == 0
    t = ts.reshape(2, 64).astype(np.int64)[0]
    return int(t[60] - t[0]), (t[62] - t[63]) / 100.0


def main():
    assert os.environ.get("AZ_PAIR_FLAGS") == "5"
    N, G, budget = 15, 512, 800
    Cs = synth.NETCONFIG2
    blob = synth.make_blob(N, 3, Cs, 0)
    for mode in ("lockstep", "persistent", "lockstep", "persistent"):
        e = capi.Engine(N, 5, G, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=0)
        e.load_weights(blob)
        e.selfplay_begin(budget)
        run = e.selfplay_run if mode == "lockstep" else e.selfplay_run_pair
        run(budget)
        e.sync()
        t0 = time.perf_counter()
        run(2 * budget)
        e.sync()
        el = time.perf_counter() - t0
        cyc, us = stamps()
        print({"mode": mode, "us_per_round": el / (2 * budget) * 1e6, "tile_cycles": cyc, "tile_us": us, "GHz": cyc / us / 1e3}, flush=True)
        e.close()


if __name__ == "__main__":
    main()
