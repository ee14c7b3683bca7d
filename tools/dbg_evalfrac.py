import sys, os
sys.path.insert(0, "/root/repo")
from alphazero_amd import capi, synth
N,G,budget=15,512,800
Cs=synth.NETCONFIG2
e=capi.Engine(N,5,G,Cs=Cs,max_budget=budget,noise_mode=capi.AZ_NOISE_DEVICE,seed=0)
e.load_weights(synth.make_blob(N,3,Cs,0))
e.selfplay_begin(budget)
prev=e.stats()
for step in range(8):
    e.selfplay_run_pair(8*budget)
    e.replay_drain(1<<15)
    st=e.stats()
    d={k:st[k]-prev[k] for k in ("evals","descents","sims_backed_up","moves","games")}
    print(step, d, "evals per slot-round %.4f" % (d["evals"]/(G*8*budget)), "sims per slot-round %.4f" % (d["sims_backed_up"]/(G*8*budget)), flush=True)
    prev=st
