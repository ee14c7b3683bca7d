#!/usr/bin/env python3
"""BASELINE configs[3]: Blokus 20x20 four players, 2048 concurrent games x 400 simulations per search, on the
device's tree kernel, with the integer-hash evaluator (env + tree path alone) or, with --tower, with the HIP
tower as the leaf evaluator.  One JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel summary."""
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
    ap.add_argument("--tower", action="store_true", help="evaluate the leaves with the HIP tower (NetConfig 2 on 20x20, 5 planes in, "
                                                         "91 policy channels, 4 values) instead of the integer-hash evaluator")
    ap.add_argument("--selfplay-moves", type=int, default=0,
                    help="afterwards play this many self-play moves of every game (search from a fresh root + move, Dirichlet "
                         "noise drawn on the device, finished games restart in place) and report them under \"selfplay\"")
    ap.add_argument("--reuse", action="store_true",
                    help="the reference's loop in full: the tree is carried from move to move (__make_root + find, mcts.hpp:385-421) and "
                         "every game starts with a warm-up search (generate.cpp:77)")
    ap.add_argument("--leaves", type=int, default=1,
                    help="descents per game and round under virtual loss (MCTS(nthreads, ...), mcts.hpp:54-72): a round evaluates "
                         "games x leaves positions")
    ap.add_argument("--precision", choices=["fp16", "bf16", "fp16x3"], default="fp16", help="operand format of the tower (--tower)")
    a = ap.parse_args()
    from alphazero_amd import capi, synth
    dev = capi.Blokus(a.games, seed=3)
    dev.search_config(max_budget=a.budget, reuse=a.reuse, leaves=a.leaves)
    dev.playout(a.pieces)
    dev.sync()
    tower = None
    if a.tower:
        Cs = synth.NETCONFIG2
        tower = capi.Engine(20, 5, a.games * a.leaves, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1,
                            precision={"fp16": capi.AZ_PREC_FP16, "bf16": capi.AZ_PREC_BF16, "fp16x3": capi.AZ_PREC_FP16X3}[a.precision])
        tower.load_weights(synth.make_blob(20, 5, Cs, 0, c_out=91, value_dim=4))
    run = (lambda b: dev.search_tower(tower, b)) if a.tower else dev.search
    run(8)                                          # warm-up
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(a.searches):
        run(a.budget)
    dev.sync()
    el = time.perf_counter() - t0
    st = dev.search_stats()
    _, count = dev.legal_mask(want_mask=False)
    sims = a.games * a.budget * a.searches
    selfplay = None
    if a.selfplay_moves > 0:
        dev.reset()
        dev.noise_device(True)
        dev.selfplay_config(ring_rows=a.games * 84, max_budget=a.budget)
        dev.selfplay_moves(a.budget, 2 if a.reuse else 1, tower_engine=tower)          # warm-up (with reuse: the games' warm-up searches, then a move)
        dev.sync()
        t1 = time.perf_counter()
        dev.selfplay_moves(a.budget, a.selfplay_moves, tower_engine=tower)
        dev.sync()
        el2 = time.perf_counter() - t1
        sp = dev.selfplay_stats()
        selfplay = {"moves_per_s": a.games * a.selfplay_moves / el2, "sims_per_s": a.games * a.selfplay_moves * a.budget / el2,
                    "seconds": el2, "moves_timed": a.games * a.selfplay_moves, "games_finished": sp["games"], "rows_in_ring": sp["rows"],
                    "rows_dropped": sp["dropped"], "tree": dev.search_stats(),
                    "note": ("from the empty board: the tree carried from move to move, %d simulations per move, device Dirichlet noise at fresh roots" if a.reuse else
                             "from the empty board: one search of %d simulations from a fresh root per move, device Dirichlet noise") % a.budget}
    print(json.dumps({"metric": "mcts_simulations_per_sec", "value": sims / el, "unit": "sims/s",
                      "config": {"workload": "blokus 20x20 4-player, %d concurrent games x %d sims per search, %s" % (
                                     a.games, a.budget, "NetConfig-2 tower (stem + 9 ResBlocks x 128 ch, 5 planes in, 91 policy channels, 4 values), "
                                     "%s operands, random-init weights" % a.precision if a.tower else "integer-hash evaluator (env + tree kernels only)"),
                                 "pieces_on_board": a.pieces, "leaves_per_round": a.leaves, "mean_legal_actions_at_root": float(count.mean())},
                      "ms_per_round": el / (a.searches * (-(-a.budget // a.leaves) + 3)) * 1e3, "searches": a.searches, "seconds": el, "tree": st,
                      **({"selfplay": selfplay} if selfplay else {})}))


if __name__ == "__main__":
    main()
