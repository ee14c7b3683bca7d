#!/usr/bin/env python3
"""Per-kernel, per-launch means of the counters collected by tools/pmc_bench.sh -> <dir>/summary.json."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"tag": tag, "workload": "bench.py --steps 1 --warmup 0 (15x15, 512 games, 800 sims/move, fp16 operands)",
           "units": "means per launch, summed over the launch's waves; SQ_INSTS_* are instruction counts, SQ_*_CYCLES / SQ_WAIT_* / "
                    "SQ_ACTIVE_* tick once per 4 shader cycles, SQ_VALU_MFMA_BUSY_CYCLES once per cycle",
           "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())}
                       for k, cs in agg.items()}}
    t = res["kernels"].get("az_tower_fused_kernel")
    if t and t.get("SQ_WAVES"):
        w = t["SQ_WAVES"]
        cyc = 4.0 * t["SQ_WAVE_CYCLES"] / w
        res["az_tower_fused_kernel_per_wave"] = {
            "wave_cycles": cyc,
            "mfma_instructions": t["SQ_INSTS_MFMA"] / w,
            "mfma_busy_frac": t["SQ_VALU_MFMA_BUSY_CYCLES"] / w / cyc,
            "wait_inst_any_frac": 4.0 * t["SQ_WAIT_INST_ANY"] / w / cyc,
            "wait_inst_lds_frac": 4.0 * t.get("SQ_WAIT_INST_LDS", 0.0) / w / cyc,
            "lds_instructions": t.get("SQ_INSTS_LDS", 0.0) / w,
            "lds_bank_conflict_frac_of_lds_active": (t.get("SQ_LDS_BANK_CONFLICT", 0.0) / t["SQ_LDS_IDX_ACTIVE"]) if t.get("SQ_LDS_IDX_ACTIVE") else None,
            "vmem_rd_instructions": t.get("SQ_INSTS_VMEM_RD", 0.0) / w,
            "valu_instructions": t.get("SQ_INSTS_VALU", 0.0) / w,
        }
    json.dump(res, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
    print(json.dumps(res.get("az_tower_fused_kernel_per_wave", {}), indent=1))


if __name__ == "__main__":
    main()
