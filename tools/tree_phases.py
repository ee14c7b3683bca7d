#!/usr/bin/env python3
"""Split of the tree kernel's time between its apply and select phases: drives the search
through the external-evaluator protocol (one launch per phase) so that a kernel trace
(`rocprofv3 --kernel-trace --output-format csv -- python3 tools/tree_phases.py`) lists
az_round_kernel dispatches alternating select, apply, select, ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import capi  # noqa: E402


def main():
    N, G, budget = 15, 512, 200
    A = N * N
    eng = capi.Engine(N, 5, G, eval_mode=capi.AZ_EVAL_EXTERNAL, max_budget=budget)
    rng = np.random.default_rng(0)
    eng.env_reset()
    eng.search_begin(budget)
    pol = rng.random((G, A)).astype(np.float32)
    pol /= pol.sum(1, keepdims=True)
    val = np.zeros((G, 2), np.float32)
    n = 0
    while True:
        planes, pending, running = eng.search_select()
        if running == 0:
            break
        v0 = rng.uniform(-0.3, 0.3, G).astype(np.float32)
        val[:, 0], val[:, 1] = v0, -v0
        eng.search_apply(np.roll(pol, n, axis=1), val)
        n += 1
    out, rn = eng.search_result()
    print("rounds", n, "root_n", int(rn[0]))


if __name__ == "__main__":
    main()
