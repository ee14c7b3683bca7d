#!/usr/bin/env python3
"""Tower-only benchmark (BASELINE.json configs[4]: 15x15, 20 blocks x 256 ch,
leaf batch 8192): evaluations/s of PVNetwork::forward on device-resident boards.

    python tools/tower_bench.py --board 15 --channels 256 --blocks 20 --batch 8192
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--precision", choices=["bf16", "fp16", "fp16x3"], default="fp16")
    a = ap.parse_args()
    import torch
    from alphazero_amd import capi, synth
    N, C, B = a.board, a.channels, a.batch
    Cs = [C] * (a.blocks + 1)
    eng = capi.Engine(N, 5, B, Cs=Cs, max_budget=1, node_cap=8,
                      precision={"fp16": capi.AZ_PREC_FP16, "bf16": capi.AZ_PREC_BF16, "fp16x3": capi.AZ_PREC_FP16X3}[a.precision])
    eng.load_weights(synth.make_blob(N, 3, Cs, 0))
    boards = torch.from_numpy(synth.random_boards(B, N, seed=0, occupancy=0.3)).cuda()
    pol = torch.empty(B, N * N, dtype=torch.float32, device="cuda")
    val = torch.empty(B, 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.tower_forward_dev(boards.data_ptr(), B, pol.data_ptr(), val.data_ptr())
    eng.sync()
    eng.profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        eng.tower_forward_dev(boards.data_ptr(), B, pol.data_ptr(), val.data_ptr())
    eng.sync()
    el = (time.perf_counter() - t0) / a.iters
    prof = eng.profile_read()
    A = N * N
    flops_eval = 2.0 * 9 * A * (3 * C + 2 * a.blocks * C * C) + 2.0 * A * C * 2 + 2.0 * (64 * A + 128)
    fused = eng.tower_is_fused()
    per_launch = (2.0 * 9 * A * C * C * B) if not fused else flops_eval * B
    out = {"metric": "tower_evals_per_sec", "value": B / el, "unit": "evals/s", "ms_per_forward": el * 1e3,
           "config": {"board": N, "channels": C, "blocks": a.blocks, "batch": B, "fused": fused, "dtype": a.precision},
           "tower_tflops": flops_eval * B / el / 1e12,
           "kernel_avg_us": prof["conv_avg_us"], "kernel_tflops": per_launch / (prof["conv_avg_us"] * 1e-6) / 1e12,
           "policy_sum_check": float(pol.sum(1).mean().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
