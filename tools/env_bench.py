#!/usr/bin/env python3
"""Env::step kernel alone (az_env_step_kernel): G games advance by one random legal move per
call.  Run under `rocprofv3 --kernel-trace --stats` for the kernel's duration; algorithmic
bytes per step = read + write of the u8 [3,N,N] board the reference's state holds (1 350 B at
15x15) + N*N/8 B legal mask + 16 B of meta."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import capi  # noqa: E402


def main():
    N, G, steps = 15, int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 60
    eng = capi.Engine(N, 5, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=1, node_cap=8)
    eng.env_reset()
    rng = np.random.default_rng(0)
    over = np.zeros(G, bool)
    order = np.stack([rng.permutation(N * N) for _ in range(G)]).astype(np.int32)   # a random legal line per game
    for s in range(steps):
        acts = np.where(over, -1, order[:, s]).astype(np.int32)
        _, done = eng.env_step(acts)
        over |= done.astype(bool)
    print("games", G, "steps", steps, "finished", int(over.sum()))


if __name__ == "__main__":
    main()
