#!/usr/bin/env python3
"""Summarise the SQ counter passes of `scripts/gpu_job.sh <tag> pmccp` (gpurun_out/<tag>/pmccp_N = the
100-tick benchmark, pmccpc_N = the crowded world) for the agent-step kernels: per kernel the mean over its
last dispatches, and the derived ratios VERDICT round 2 asked for (LDS share, bank conflicts, waits).
Usage: cp_counters.py <tag> [out.json]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench    # noqa: E402

KERNELS = ("k_cp_rows", "k_cp_small", "k_cp_heavy", "k_agent_nbr", "k_agent_mid", "k_cohesion", "k_field_bfs",
           "k_agent_full", "k_sp_place")


def short(name):
    return name.split("(")[0].replace("void ", "").strip().split("<")[0]


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for sub in sorted(os.listdir(d)):
        p = os.path.join(d, sub, "p_counter_collection.csv")
        if not os.path.exists(p):
            continue
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if k in KERNELS:
                acc[(sub.split("_")[0], k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = {"vgpr": int(r["VGPR_Count"]), "lds_block": int(r["LDS_Block_Size"]),
                           "wg": int(r["Workgroup_Size"])}
    return acc, meta


def main():
    tag = sys.argv[1]
    acc, meta = load(os.path.join(ROOT, "gpurun_out", tag))
    out = {"source": "rocprofv3 --kernel-trace --pmc <4 SQ counters per pass>, three passes each of bench.py --steps 100 "
                     "--warmup 5 (pmccp: mean of each kernel's last 6 ticks' dispatches = ticks ~105-110) and bench.py "
                     "--crowded --steps 20 --warmup 3 (pmccpc: the dispatches after the first 3 ticks)",
           "csrc_sha": bench.csrc_sha(), "files": bench.csrc_files(), "kernels": {}}
    for (run, k), cs in sorted(acc.items()):
        n = max(len(v) for v in cs.values())
        per_tick = 2 if k == "k_cp_rows" else 1
        take = 6 * per_tick if run == "pmccp" else max(1, n - 3 * per_tick)
        d = {c: sum(v[-take:]) / len(v[-take:]) * per_tick for c, v in cs.items()}
        wc = d.get("SQ_WAVE_CYCLES")
        r = dict(d)
        if wc:
            for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_INST_CYCLES_VMEM"):
                if c in d:
                    r[c + "/WAVE_CYCLES"] = d[c] / wc
        if d.get("SQ_INSTS_VALU"):
            for c in ("SQ_INSTS_LDS", "SQ_INSTS_SALU"):
                if c in d:
                    r[c + "/INSTS_VALU"] = d[c] / d["SQ_INSTS_VALU"]
        if d.get("SQ_INSTS_LDS"):
            r["LDS_BANK_CONFLICT_cycles_per_LDS_inst"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_INSTS_LDS"]
        r.update(meta.get(k, {}))
        out["kernels"].setdefault(k, {})["bench_tick_105" if run == "pmccp" else "crowded_world"] = r
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    for k, v in out["kernels"].items():
        for run, r in v.items():
            print("%-12s %-15s vgpr %3s  " % (k, run, r.get("vgpr")) + "  ".join(
                "%s %.3g" % (c.replace("SQ_", ""), r[c]) for c in sorted(r) if c not in ("vgpr", "lds_block", "wg")))


if __name__ == "__main__":
    main()
