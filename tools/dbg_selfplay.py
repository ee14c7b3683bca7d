import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import azo
from alphazero_amd import capi
N, K, budget, G = 5, 4, 48, 4
A = N * N
seed = 77
eng = capi.Engine(N, K, G, eval_mode=capi.AZ_EVAL_HASH, max_budget=budget, seed=seed)
noise = np.stack([azo.dirichlet(500 + g, A)[0] for g in range(G)])
eng.set_noise(noise)
eng.selfplay_begin(budget)
eng.selfplay_run((2 * budget + 1 + (A - 1) * budget) * 2 + 8)
cells, tm, pol, rew = eng.replay_drain()
print(eng.stats())
# split device rows into games
games = []
for i in range(len(tm)):
    if cells[i].sum() == 0:
        games.append([])
    games[-1].append(i)
print("device games:", [(len(g), rew[g[0]].tolist()) for g in games])
for g in range(G):
    m = azo.MCTS(N, K, 3.0, 3.0, "hash")
    avg, gc, sc = 25.0, 0, 0
    for k in range(2):
        r = m.selfplay_game(budget, seed, g, noise[g], avg, gc, sc)
        avg, gc, sc = r["avg_len"], r["game_count"], r["sample_counter"]
        print("oracle slot", g, "game", k, "len", r["n_moves"], "reward", r["reward"].tolist(), "actions", r["actions"].tolist(), "avg", avg, "sc", sc)
        # find matching device game by first-row policy and then compare move by move
        for dg in games:
            if np.array_equal(pol[dg[0]], r["policy"][0]):
                # compare
                n = min(len(dg), r["n_moves"])
                first_bad = None
                for i in range(n):
                    if not (np.array_equal(cells[dg[i]], r["cells"][i]) and np.array_equal(pol[dg[i]], r["policy"][i])):
                        first_bad = i; break
                print("   device game len", len(dg), "rew", rew[dg[0]].tolist(), "first mismatch at move", first_bad)
                if first_bad is not None:
                    i = first_bad
                    print("   cells dev", cells[dg[i]].tolist()); print("   cells orc", r["cells"][i].tolist())
                    print("   pol eq", np.array_equal(pol[dg[i]], r["policy"][i]))
