#!/usr/bin/env python3
"""Self-play hot-path benchmark: MCTS simulations/s, whole job.

One step = one move of every game = `sims/move` passes of the hot path
(env step -> PUCT select -> tower evaluation -> expand -> backup) over G
concurrent 15x15 gomoku games per GPU, including the re-root that carries every
tree over to the next move (BASELINE.json configs[2] sharded by game: 4096
games over 8 GPUs = 512 per GPU; at N=1 the same 512-game shard).  A step runs
as ONE launch of the persistent tower / tree pair (az_selfplay_run_pair); the
same games as lockstep launches are measured behind it, informationally (and
are the headline with --lockstep, several contexts, --leaves > 1, or ranks that
share a GPU).  Weak scaling: every rank owns its own games, there is no
collective inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak of one MI355X
HBM_PEAK_GBS = 8000.0                # HBM3E peak of one MI355X


def cpu_baseline(seconds, N, K, budget, nthreads):
    """The reference's own MCTS/Evaluator/PVNetwork (compiled unmodified into
    oracle/_ref/az_ref) running the generate.cpp loop on this host's cores."""
    ref = os.path.join(ROOT, "oracle", "_ref", "az_ref")
    if os.path.exists(ref):
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        with tempfile.TemporaryDirectory() as d:
            try:
                out = subprocess.run([ref, "bench", str(N), str(K), str(nthreads), str(budget), str(seconds), d],
                                     env=env, capture_output=True, text=True, timeout=seconds * 2 + 60).stdout
                line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
                r = json.loads(line)
                return {"value": r["sims_per_s"], "unit": "sims/s", "cores": int(r["torch_threads"]), "kind": "reference",
                        "sample": "reference mcts.hpp+evaluator.hpp+network.hpp (libtorch CPU, NetConfig 2), generate.cpp loop, "
                                  "%dx%d gomoku, nthreads=%d, batch 16, %d sims/move, %.0f s wall, %d sims, %d torch intra-op threads"
                                  % (N, N, nthreads, budget, r["seconds"], r["sims"], r["torch_threads"])}
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("reference baseline failed (%r); falling back to the oracle port\n" % (e,))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import azo
    from alphazero_amd import synth
    Cs = synth.NETCONFIG2
    tower = azo.Tower(N, 3, Cs, synth.make_blob(N, 3, Cs, 0))
    m = azo.MCTS(N, K, 3.0, 3.0, "tower", tower)
    t0 = time.time()
    sims = 0
    while time.time() - t0 < seconds:
        m.search([], 4, None)
        sims += 4
    el = time.time() - t0
    return {"value": sims / el, "unit": "sims/s", "cores": 1, "kind": "port",
            "sample": "scalar C++ restatement, %dx%d, %d sims in %.1f s" % (N, N, sims, el)}


def other_configs(device):
    """Short legs of the other BASELINE.json configurations, after the headline and outside its timed region, so
    that one record covers every config: each {workload, value, unit, roofline} with the dominant kernel's HIP-event
    launch time and FLOPs of real board cells per launch."""
    from alphazero_amd import capi, synth
    import torch
    out = []
    Cs = synth.NETCONFIG2

    def roof(kernel, flops, prof, hbm_bytes=None):
        us = prof["conv_avg_us"]
        tf = flops / (us * 1e-6) / 1e12 if us > 0 else 0.0
        r = {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": tf / MFMA_BF16_PEAK_TFLOPS, "avg_launch_us": us, "launches_sampled": prof["conv_launches"],
             "flops_per_launch": flops, "traffic": None}
        if hbm_bytes is not None and us > 0:
            r["algorithmic_hbm_gbs"] = hbm_bytes / (us * 1e-6) / 1e9
            r["algorithmic_hbm_frac"] = r["algorithmic_hbm_gbs"] / HBM_PEAK_GBS
        return r

    # configs[1]: gomoku 9x9, 256 games x 400 sims/move, self-play with the fused tower in 128-row tiles, every move's
    # rounds as one launch of the persistent pair with chained terminal simulations (as the headline).  With 256 games
    # every CU holds ONE game, so a game's tree step cannot hide under another game's tile here: what the pair saves
    # is the launch boundaries, what the chain saves is the evaluations of simulations that need none.
    N, G, budget, moves = 9, 256, 400, 30
    eng = capi.Engine(N, 5, G, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=0, device=device)
    eng.load_weights(synth.make_blob(N, 3, Cs, 0))
    eng.selfplay_chain(6)
    eng.selfplay_begin(budget)
    for _ in range(3):
        eng.selfplay_run_pair(budget)
    eng.profile_enable(1)
    eng.sync()
    s0 = eng.stats()["sims_backed_up"]
    t0 = time.perf_counter()
    for _ in range(moves):
        eng.selfplay_run_pair(budget)
    eng.sync()
    el = time.perf_counter() - t0
    st = eng.stats()
    pr = eng.pair_read()
    A, C = N * N, Cs[0]
    prof = {"conv_avg_us": pr["launch_avg_us"], "conv_launches": pr["launches"]}
    out.append({"config": 1, "workload": "gomoku 9x9 self-play, 256 concurrent games, 400 evaluations per game per move as one launch of the "
                                         "persistent pair (chain 6), NetConfig-2 tower, fp16 operands",
                "value": (st["sims_backed_up"] - s0) / el, "unit": "sims/s", "seconds": el, "ms_per_round": el / (moves * budget) * 1e3,
                "sims_per_game_round": (st["sims_backed_up"] - s0) / (moves * budget * G),
                "roofline": roof("az_tower_pair_kernel (%d rounds per launch, hand-over waits and the serial tree step included)" % budget,
                                 (2.0 * 9 * A * C * C * 2 * (len(Cs) - 1) + 2.0 * 27 * A * C) * G * budget, prof),
                "tree_step_avg_us": pr["tree_step_avg_us"], "error_games": st["error_games"]})
    eng.close()

    # configs[3]: Blokus 20x20 four players, 2048 games x 400 sims per search, the tower as evaluator
    G, budget, searches = 2048, 400, 2
    dev = capi.Blokus(G, device=device, seed=3)
    dev.search_config(max_budget=budget)
    dev.playout(9)
    dev.sync()
    tower = capi.Engine(20, 5, G, Cs=Cs, c_in=5, c_out=91, value_dim=4, max_budget=1, device=device)
    tower.load_weights(synth.make_blob(20, 5, Cs, 0, c_out=91, value_dim=4))
    dev.search_tower(tower, 8)
    dev.sync()
    tower.profile_enable(max(1, searches * (budget + 2) * 2 * (len(Cs) - 1) // 3600))
    t0 = time.perf_counter()
    for _ in range(searches):
        dev.search_tower(tower, budget)
    dev.sync()
    el = time.perf_counter() - t0
    prof = tower.profile_read()
    bst = dev.search_stats()
    rows = G * 441.0                                   # raster rows of a launch (21 x 21 per board)
    out.append({"config": 3, "workload": "blokus 20x20 4-player, 2048 concurrent games x 400 sims per search, NetConfig-2 tower "
                                         "(5 planes in, 91 policy channels, 4 values) as evaluator, fp16 operands",
                "value": G * budget * searches / el, "unit": "sims/s", "seconds": el, "ms_per_round": el / (searches * (budget + 2)) * 1e3,
                # mean over a block's two convolutions: the first moves 2 + 2 B per activation element through HBM, the
                # second 2 + 4 + 4 + 2 (16-bit in, fp32 residual in and out, 16-bit copy out)
                "roofline": roof("az_conv3x3_t256s_kernel (mean of a ResBlock's two convolutions)", 2.0 * 9 * 400 * C * C * G, prof,
                                 hbm_bytes=rows * C * 8.0),
                "tree_restarts": bst["restarts"]})
    dev.close()
    tower.close()

    # configs[4]: tower only, 20 ResBlocks x 256 channels, 15x15, batch 8192; fp16 operands and the compensated mode
    N, C4, blocks, B = 15, 256, 20, 8192
    Cs4 = [C4] * (blocks + 1)
    boards = torch.from_numpy(synth.random_boards(B, N, seed=0, occupancy=0.3)).cuda(device)
    pol = torch.empty(B, N * N, dtype=torch.float32, device="cuda:%d" % device)
    val = torch.empty(B, 2, dtype=torch.float32, device="cuda:%d" % device)
    torch.cuda.synchronize()
    blob4 = synth.make_blob(N, 3, Cs4, 0)
    A = N * N
    flops_eval = 2.0 * 9 * A * (3 * C4 + 2 * blocks * C4 * C4) + 2.0 * A * C4 * 2 + 2.0 * (64 * A + 128)
    for prec, name, iters in ((capi.AZ_PREC_FP16, "fp16", 5), (capi.AZ_PREC_FP16X3, "fp16x3", 2)):
        eng = capi.Engine(N, 5, B, Cs=Cs4, max_budget=1, node_cap=8, precision=prec, device=device)
        eng.load_weights(blob4)
        eng.tower_forward_dev(boards.data_ptr(), B, pol.data_ptr(), val.data_ptr())
        eng.sync()
        eng.profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.tower_forward_dev(boards.data_ptr(), B, pol.data_ptr(), val.data_ptr())
        eng.sync()
        el = (time.perf_counter() - t0) / iters
        prof = eng.profile_read()
        e = {"config": 4, "workload": "tower only: 15x15, 20 ResBlocks x 256 ch, leaf batch 8192, %s" % (
                 "fp16 operands" if name == "fp16" else "AZ_PREC_FP16X3 (hi + lo fp16 operands, reduction over 3x the channels: the parity mode)"),
             "value": B / el, "unit": "evals/s", "ms_per_forward": el * 1e3, "tower_tflops": flops_eval * B / el / 1e12, "dtype": name}
        if name == "fp16":
            e["roofline"] = roof("az_conv3x3_t256s_kernel (mean of a ResBlock's two convolutions)", 2.0 * 9 * A * C4 * C4 * B, prof,
                                 hbm_bytes=B * 256.0 * C4 * 8.0)
        else:
            e["roofline"] = roof("az_conv3x3_t256s_kernel, Cin = 3 x 256 (MFMA work of three convolutions per launch)",
                                 3 * 2.0 * 9 * A * C4 * C4 * B, prof)
            e["roofline"]["note"] = "achieved counts the MFMA work issued (three products per weight-activation pair); a third of it is the convolution"
        out.append(e)
        eng.close()

    # the headline workload in the parity mode (per-layer kernels, every convolution over 3x the channels)
    N, G, budget = 15, 512, 800
    eng = capi.Engine(N, 5, G, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=0, device=device,
                      precision=capi.AZ_PREC_FP16X3)
    eng.load_weights(synth.make_blob(N, 3, Cs, 0))
    eng.selfplay_begin(budget)
    eng.selfplay_run(200)
    eng.sync()
    s0 = eng.stats()["sims_backed_up"]
    t0 = time.perf_counter()
    eng.selfplay_run(1600)
    eng.sync()
    el = time.perf_counter() - t0
    out.append({"config": 2, "workload": "the headline workload (gomoku 15x15, 512 games, 800 sims/move) in AZ_PREC_FP16X3, the parity mode",
                "value": (eng.stats()["sims_backed_up"] - s0) / el, "unit": "sims/s", "seconds": el, "ms_per_round": el / 1600 * 1e3, "dtype": "fp16x3"})
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; one step = one move of every game = sims/move rounds")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before the timed region")
    ap.add_argument("--games", type=int, default=512, help="concurrent games per GPU")
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--budget", type=int, default=800)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["bf16", "fp16", "fp16x3"], default="fp16",
                    help="operand format of the tower's convolutions: fp16 (default, the throughput mode), bf16, or fp16x3 "
                         "(compensated: hi + lo fp16 operands, fp32-class agreement with the reference network, per-layer kernels)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs of BASELINE configs[1], [3], [4] that the default single-GPU run appends as other_configs")
    ap.add_argument("--contexts", type=int, default=1,
                    help="split the GPU's games over this many engine contexts (own stream each)")
    ap.add_argument("--leaves", type=int, default=1,
                    help="descents per tree per round (1 = the reference at nthreads=1, bit-exact; >1 = its nthreads=L "
                         "in a fixed order, evaluation batch = games x leaves)")
    ap.add_argument("--lockstep", action="store_true",
                    help="time the rounds as lockstep launches (evaluator, tree round, ...) instead of one launch of the "
                         "persistent tower / tree pair per step")
    ap.add_argument("--chain", type=int, default=6,
                    help="persistent pair only: simulations ending in a stored terminal leaf (which call no network, mcts.hpp:232-233) "
                         "that a game may run back to back inside one round, so that the round's evaluation is not spent on a "
                         "stale board (az_selfplay_chain); 0 = one simulation per game per round, as the lockstep launches")
    ap.add_argument("--no-lockstep-probe", action="store_true",
                    help="skip the second, informational measurement of the same games as lockstep launches")
    ap.add_argument("--full-games", type=float, default=40.0,
                    help="after the timed region keep playing for this many seconds and report finished games/hour")
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=None,
                    help="preset of BASELINE.json configs[i]: 1 = 9x9, 256 games, 400 sims/move; 2 = the default "
                         "(15x15, 512 games per GPU, 800 sims/move); 3 = blokus 20x20 4-player, 2048 games x 400 sims with "
                         "the tower as evaluator, three searches then three self-play moves (runs tools/blokus_search_bench.py, single GPU); 4 = tower only, "
                         "20 blocks x 256 channels, batch 8192 (runs tools/tower_bench.py, single GPU)")
    args = ap.parse_args()
    if args.config == 1:
        args.board, args.games, args.budget = 9, 256, 400
    elif args.config == 3:
        sys.argv = [os.path.join(ROOT, "tools", "blokus_search_bench.py"), "--tower", "--selfplay-moves", "3", "--reuse"]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import blokus_search_bench
        return blokus_search_bench.main()
    elif args.config == 4:
        sys.argv = [os.path.join(ROOT, "tools", "tower_bench.py"), "--precision", args.precision]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import tower_bench
        return tower_bench.main()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from alphazero_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # AZ_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks: ranks share the devices
    # there are and the collectives (a weight broadcast, a few scalars) run on host tensors
    backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    cdev = "cuda" if backend == "nccl" else "cpu"          # where the collectives' tensors live
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N, K, G, budget = args.board, 5, args.games, args.budget
    Cs = synth.NETCONFIG2
    prec = {"fp16": capi.AZ_PREC_FP16, "bf16": capi.AZ_PREC_BF16, "fp16x3": capi.AZ_PREC_FP16X3}[args.precision]
    NC = args.contexts
    assert G % NC == 0
    # weights: rank 0 builds the synthetic blob, RCCL broadcasts it (the only collective of this path)
    blob = wt = None
    if use_dist:
        nw = synth.blob_size(N, 3, Cs) if hasattr(synth, "blob_size") else len(synth.make_blob(N, 3, Cs, 0))
        wt = torch.empty(nw, dtype=torch.float32, device=cdev)
        if rank == 0:
            wt.copy_(torch.from_numpy(synth.make_blob(N, 3, Cs, 0)))
        dist.broadcast(wt, 0)
        wt = wt.cuda()
        torch.cuda.synchronize()
    else:
        blob = synth.make_blob(N, 3, Cs, 0)

    def make_engines(chain):
        es = [capi.Engine(N, K, G // NC, Cs=Cs, max_budget=budget, noise_mode=capi.AZ_NOISE_DEVICE, seed=0,
                          device=local_rank, game_slot_base=rank * G + i * (G // NC), precision=prec,
                          leaves_per_round=args.leaves) for i in range(NC)]
        for e in es:
            if use_dist:
                e.load_weights_dev(wt.data_ptr(), e.weights_size())
            else:
                e.load_weights(blob)
            if args.leaves == 1:
                e.selfplay_chain(chain)
        return es

    # one step = one move's worth of rounds (sims/move descents per game, L per round)
    rps = (budget + args.leaves - 1) // args.leaves
    TALLY = ("descents", "tree_levels", "children_scanned", "children_expanded", "evals")

    def agg(sts):
        return {k: sum(x[k] for x in sts) if k not in ("max_nodes", "max_edges") else max(x[k] for x in sts) for k in sts[0]}

    def timed_leg(pair, chain, steps):
        """warm-up + the timed region, either with every step as ONE launch of the persistent tower / tree pair
        (az_selfplay_run_pair) or as lockstep launches (az_selfplay_run: evaluator, tree round, evaluator, ...).
        chain: simulations that end in a stored terminal leaf (no evaluation, mcts.hpp:232-233) a game may run back
        to back inside one round (az_selfplay_chain)."""
        es = make_engines(chain)

        def run(rounds):
            if pair:
                for done in range(0, rounds, rps):
                    es[0].selfplay_run_pair(min(rps, rounds - done))
                return
            # launches are asynchronous: interleave the contexts in slices so every stream stays fed
            done = 0
            while done < rounds:
                k = min(16, rounds - done)
                for e in es:
                    e.selfplay_run(k)
                done += k

        def barrier(engines=True):
            for e in (es if engines else []):
                e.sync()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for e in es:
            e.selfplay_begin(budget)
        # the pair needs its two kernels resident together; a rank that does not get that says so here, and then
        # every rank leaves the leg together (the collectives below stay matched)
        ok, why = 1, ""
        try:
            run(args.warmup * rps)
            for e in es:
                e.sync()
        except capi.AzError as ex:
            if not pair:
                raise
            ok, why = 0, str(ex)
        if use_dist and pair:
            t = torch.tensor([ok], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0 and ok:
                ok, why = 0, "another rank's persistent pair did not run"
        if not ok:
            for e in es:
                e.close()
            return {"error": why}
        # HIP events around every `every`-th evaluator call and tree round of the timed region, within the
        # library's event pool (4096 spans): every launch when the region is short (every launch of the pair)
        spans_per_round = 3 if Cs[0] <= 128 and N <= 15 and prec != capi.AZ_PREC_FP16X3 else 2 * (len(Cs) - 1) + 2
        every = 1 if pair else max(1, -(-(steps * rps * spans_per_round) // 3600))
        es[0].profile_enable(every)
        barrier()
        st0 = agg([e.stats() for e in es])
        t0 = time.perf_counter()
        # (a pair that stops being resident together in the timed region -- another process arriving on the GPU --
        # ends the leg the same way as one that never was: every rank leaves together, the lockstep form takes over)
        ok, why = 1, ""
        try:
            run(steps * rps)
            for e in es:
                e.sync()
        except capi.AzError as ex:
            if not pair:
                raise
            ok, why = 0, str(ex)
        barrier(engines=bool(ok))
        el = time.perf_counter() - t0
        if use_dist and pair:
            t = torch.tensor([ok], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0 and ok:
                ok, why = 0, "another rank's persistent pair did not run"
        if not ok:
            for e in es:
                e.close()
            return {"error": why}
        st = agg([e.stats() for e in es])
        leg = {"pair": pair, "chain": chain, "steps": steps, "engs": es, "run": run, "el_local": el, "el": el, "every": every, "st": st,
               "sims_local": st["sims_backed_up"] - st0["sims_backed_up"],
               "tally": {k: st[k] - st0[k] for k in TALLY},
               "prof": es[0].pair_read() if pair else es[0].profile_read(),
               "fused": es[0].tower_is_fused(), "batch": es[0].selfplay_batch()}
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            leg["el"] = float(t.item())
            t = torch.tensor([leg["sims_local"]], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            leg["sims_all"] = int(t.item())
            # what each rank contributed: its own device counter and its own clock (value = sum / max)
            mine = torch.tensor([float(leg["sims_local"]), el], dtype=torch.float64, device=cdev)
            allr = [torch.zeros(2, dtype=torch.float64, device=cdev) for _ in range(world)]
            dist.all_gather(allr, mine)
            leg["per_rank"] = {"sims": [int(x[0].item()) for x in allr], "seconds": [float(x[1].item()) for x in allr]}
        else:
            leg["sims_all"] = leg["sims_local"]
            leg["per_rank"] = None
        if st["error_games"]:
            raise SystemExit("tree arena overflow in %d games" % st["error_games"])
        return leg

    # The headline runs every step as one launch of the persistent pair wherever that form exists (the one-launch
    # tower, one leaf per round, one context) and each rank has its GPU to itself; --lockstep, several contexts,
    # --leaves > 1 or ranks sharing a GPU (the gloo rehearsal) take the lockstep launches.
    probe = capi.Engine(N, K, 1, Cs=Cs, max_budget=2, precision=prec, device=local_rank)
    fused_tower = probe.tower_is_fused()
    probe.close()
    want_pair = fused_tower and NC == 1 and args.leaves == 1 and not args.lockstep and backend != "gloo"
    pair_fallback = None
    leg = timed_leg(True, args.chain, args.steps) if want_pair else None
    if leg is not None and "error" in leg:
        pair_fallback = leg["error"]
        leg = None
    if leg is None:
        leg = timed_leg(False, 0, args.steps)
    engs, run, el, st = leg["engs"], leg["run"], leg["el"], leg["st"]
    sims_all, per_rank, prof, tally = leg["sims_all"], leg["per_rank"], leg["prof"], leg["tally"]
    fused, batch, every = leg["fused"], leg["batch"], leg["every"]
    games_info = None
    if args.full_games > 0:
        # whole games: keep the same engine running, count games finished per wall second
        engs[0].profile_enable(0)
        t1 = time.perf_counter()
        g0, m0, s0fg = st["games"], st["moves"], st["sims_backed_up"]
        while time.perf_counter() - t1 < args.full_games:
            run(budget)
            for e in engs:
                e.replay_drain(1 << 15)
        el2 = time.perf_counter() - t1
        st2 = agg([e.stats() for e in engs])
        ng, nm, ns = st2["games"] - g0, st2["moves"] - m0, st2["sims_backed_up"] - s0fg
        if use_dist:                                       # whole-job totals
            t = torch.tensor([ng, nm, ns], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ng, nm, ns = int(t[0].item()), int(t[1].item()), int(t[2].item())
            t = torch.tensor([el2], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        games_info = {"seconds": el2, "games": ng, "moves": nm,
                      "games_per_hour": ng / el2 * 3600.0, "sims_per_s": ns / el2,
                      # finished games only: a lower bound on the rate while the first games are still running
                      "moves_per_s": nm / el2, "mean_game_length": nm / ng if ng else None,
                      "max_tree_nodes": st2["max_nodes"], "error_games": st2["error_games"],
                      "tree_restarts": st2["tree_restarts"]}
    for e in engs:
        e.close()
    engs = []

    # Informational second measurement, never `value`: the same games (same seeds, same number of timed rounds, same
    # device, same call) as lockstep launches.  It prices the tree round as a kernel of its own (`tree_roofline`)
    # and says what the persistent pair buys.
    lock = plain_pair = None
    if leg["pair"] and not args.no_lockstep_probe:
        lock = timed_leg(False, 0, args.steps)
        for e in lock["engs"]:
            e.close()
        if args.chain > 0:
            # and the pair with one simulation per game per round, a third of the way between the two: what the
            # overlap alone buys, without the chained terminal simulations
            plain_pair = timed_leg(True, 0, min(args.steps, 8))
            if "error" not in plain_pair:
                for e in plain_pair["engs"]:
                    e.close()

    if rank == 0:
        A = N * N
        C = Cs[0]
        sims = sims_all                                    # counted on the device, one per backup
        B = batch * NC                      # boards per evaluation launch (x NC: divided again below)
        conv_flops = 2.0 * 9 * A * C * C * B               # one 3x3 conv over B boards, real cells only
        kernel = "az_conv3x3_t256_kernel"
        if fused:                           # stem + all 2*num_res convs in one launch
            conv_flops = (conv_flops * 2 * (len(Cs) - 1) + 2.0 * 27 * A * C * B) / NC   # per launch of one context
            kernel = "az_tower_fused_kernel"
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of the lockstep form of this command
        # (tools/profile_bench.sh; counter collection serialises launches, which the pair cannot run under); only
        # valid for the workload they were collected on.  A pair launch walks the same tiles `rps` times.
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and (N, G, budget, NC, args.leaves) == (15, 512, 800, 1, 1):
            try:
                traffic = json.load(open(tj))["kernels"][kernel]["traffic_bytes_per_launch"]
            except Exception:  # noqa: BLE001
                traffic = None

        def tower_roofline(us, launches, flops, name, tr):
            tf = flops / (us * 1e-6) / 1e12 if us > 0 else 0.0
            return {"bound": "mfma", "kernel": name, "achieved": tf,
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                    "traffic": tr, "avg_launch_us": us, "launches_sampled": launches, "flops_per_launch": flops,
                    # informational: a bare MFMA loop at the issue floor reaches this on random bf16 operands
                    # (the chip lowers its clock under MFMA load; tools/microbench/mfma_power.hip); `peak` and
                    # `frac` stay priced against the datasheet's dense peak
                    "bare_mfma_loop_on_random_data_tflops": 1830.0}

        def tree_roofline(lg):
            # Tree kernel against the HBM roofline: algorithmic bytes = 12 B (p, n, v) per child edge read at
            # every level of every descent + 24 B per level (virtual-loss write + backup read-modify-write) +
            # 16 B per child edge written by an expansion + per evaluated leaf its planes (3A B) out and its
            # policy and value (4A + 8 B) in; counted on the device (az_stats), rank 0's games, per launch.
            tl, pf = lg["tally"], lg["prof"]
            if not (tl["descents"] > 0 and pf["tree_avg_us"] > 0):
                return None
            tree_bytes = (12.0 * tl["children_scanned"] + 24.0 * tl["tree_levels"] + 16.0 * tl["children_expanded"]
                          + tl["evals"] * (7.0 * A + 8.0)) / (lg["steps"] * rps) / NC
            tree_traffic = None
            if traffic is not None:
                try:
                    tree_traffic = json.load(open(tj))["kernels"]["az_round_kernel"]["traffic_bytes_per_launch"]
                except Exception:  # noqa: BLE001
                    tree_traffic = None
            # a span of two event records over-states a short kernel: the empty spans taken right behind the sampled
            # rounds say by how much (rocprofv3's kernel duration agrees with the difference)
            tree_us = max(pf["tree_avg_us"] - pf["empty_span_us"], 1e-3)
            gbs = tree_bytes / (tree_us * 1e-6) / 1e9
            return {"bound": "hbm", "kernel": "az_round_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": tree_traffic, "avg_launch_us": tree_us,
                    "event_span_us": pf["tree_avg_us"], "empty_event_span_us": pf["empty_span_us"],
                    "launches_sampled": pf["tree_launches"], "bytes_per_launch": tree_bytes,
                    "mean_depth": tl["tree_levels"] / tl["descents"],
                    "mean_children_per_level": tl["children_scanned"] / max(1, tl["tree_levels"]),
                    "note": "a chain of dependent loads, one per tree level: latency-bound, 512 one-wave blocks"}

        if leg["pair"]:
            step = ("%d rounds = %d evaluations per game, as ONE launch of the persistent tower / tree pair (az_tower_pair_kernel beside "
                    "az_round_pair_kernel): a move's worth (%d sims/move) and, because up to %d simulations that end in a stored terminal leaf "
                    "and call no network ride along in a round, %.3f simulations per game and round here; re-roots included"
                    % (rps, rps, budget, leg["chain"], sims / (args.steps * rps * G * world)))
            roof = tower_roofline(prof["launch_avg_us"], prof["launches"], conv_flops * rps, "az_tower_pair_kernel",
                                  traffic * rps if traffic is not None else None)
            roof["rounds_per_launch"] = rps
            roof["tree_step_avg_us"] = prof["tree_step_avg_us"]
            roof["note"] = ("a launch walks every game's tile %d times and ends when the last game's last evaluation is out: its duration "
                            "includes the hand-over waits, so `frac` is the whole round's MFMA fraction; a game's tree step (tree_step_avg_us, "
                            "timed inside the kernel) runs under another game's tile" % rps)
        else:
            step = "one move of every game: %d lockstep rounds (%d sims/move, %d per round), re-root included" % (rps, budget, args.leaves)
            roof = tower_roofline(prof["conv_avg_us"], prof["conv_launches"], conv_flops, kernel, traffic)
            roof["tree_round_avg_us"] = max(prof["tree_avg_us"] - prof["empty_span_us"], 0.0)
        line = {
            "metric": "mcts_simulations_per_sec", "value": sims / el, "unit": "sims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "step": step, "rounds_per_step": rps, "ms_per_round": el / (args.steps * rps) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "gomoku %dx%d self-play, %d concurrent games per GPU, %d sims/move, NetConfig-2 tower "
                                   "(stem + 9 ResBlocks x 128 ch), vl=3 c_puct=3" % (N, N, G, budget),
                       "games_per_gpu": G, "sims_per_move": budget, "board": N, "parallelism": "games sharded x%d" % world,
                       "contexts_per_gpu": NC, "leaves_per_round": args.leaves,
                       "rounds_as": "persistent pair" if leg["pair"] else "lockstep launches", "terminal_chain": leg["chain"]},
            "moves_per_s": sims / el / budget,
            "roofline": roof,
        }
        if pair_fallback is not None:
            line["persistent_pair_not_used"] = pair_fallback
        if not leg["pair"]:
            tr = tree_roofline(leg)
            if tr is not None:
                line["tree_roofline"] = tr
        elif lock is not None:
            # the same games as lockstep launches: the tree round as a kernel of its own, and the A/B
            tr = tree_roofline(lock)
            if tr is not None:
                line["tree_roofline"] = tr
            lp = lock["prof"]
            lr = tower_roofline(lp["conv_avg_us"], lp["conv_launches"], conv_flops, kernel, traffic)
            line["lockstep_launches"] = {
                "value": lock["sims_all"] / lock["el"], "unit": "sims/s", "ms_per_round": lock["el"] / (lock["steps"] * rps) * 1e3,
                "sims_per_game_round": lock["sims_all"] / (lock["steps"] * rps * G * world),
                "pair_over_lockstep": (sims / el) / (lock["sims_all"] / lock["el"]),
                "roofline": lr, "tree_round_avg_us": max(lp["tree_avg_us"] - lp["empty_span_us"], 0.0),
                "note": "informational: the same games, seeds and number of timed rounds on the same device in the same call, every round "
                        "as two launches on one stream (az_tower_fused_kernel, az_round_kernel); `tree_roofline` is taken here"}
        if plain_pair is not None:
            if "error" in plain_pair:
                line["pair_one_simulation_per_round"] = {"error": plain_pair["error"]}
            else:
                line["pair_one_simulation_per_round"] = {
                    "value": plain_pair["sims_all"] / plain_pair["el"], "unit": "sims/s", "steps": plain_pair["steps"],
                    "ms_per_round": plain_pair["el"] / (plain_pair["steps"] * rps) * 1e3,
                    "note": "informational: the persistent pair with az_selfplay_chain(0); against lockstep_launches this is what hiding "
                            "the tree work and the launch boundaries buys, the rest of the headline's margin is the terminal simulations "
                            "that no longer cost an evaluation"}
        line["consistency"] = {"timed_region_s": el, "rounds_timed": args.steps * rps, "profile_every": every,
                               "sims_per_s_of_full_games_phase": games_info["sims_per_s"] if games_info else None}
        if per_rank is not None:
            line["per_rank"] = per_rank
        line["full_games"] = games_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, N, K, budget, 16)
        if world == 1 and not args.no_other_configs and args.config is None and (N, G, budget, NC, args.leaves) == (15, 512, 800, 1, 1):
            try:
                line["other_configs"] = other_configs(local_rank)
            except Exception as ex:  # noqa: BLE001  (the headline stands on its own)
                line["other_configs"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
