#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_bench.sh into the committed
summaries: gpurun_out/prof_<tag>/{kernel_stats.csv,traffic.json} (copy them
into profiles/)."""
import collections
import csv
import glob
import json
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
ks = glob.glob(out + "/trace/*/*kernel_stats.csv")[0]
shutil.copy(ks, out + "/%s_kernel_stats.csv" % tag)
print(open(ks).read())
res = {"tag": tag, "workload": "bench.py defaults: gomoku 15x15, 512 games, 800 sims/move, NetConfig 2",
       "note": "KB per launch as rocprofv3 reports them; on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide "
               "coalesced read, so bytes fetched = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B stores",
       "kernels": {}}
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(out + "/%s/*/*counter_collection.csv" % sub)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k] += float(r["Counter_Value"])
        cnt[k] += 1
    for k in agg:
        if k.startswith("az_"):
            res["kernels"].setdefault(k, {})[name + "_KB"] = agg[k] / cnt[k]
for k, v in res["kernels"].items():
    v["traffic_bytes_per_launch"] = (2.0 * v.get("FETCH_SIZE_KB", 0.0) + v.get("WRITE_SIZE_KB", 0.0)) * 1024.0
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
