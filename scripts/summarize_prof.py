#!/usr/bin/env python3
"""Turn the outputs of scripts/prof_job.sh (gpurun_out/<tag>/) into the committed summaries under
profiles/: kernel stats csv (from the rocprofv3 results db), traffic.json (PMC FETCH/WRITE passes),
SQ counter summary, bench line, fuzz and pytest logs.
Usage: summarize_prof.py <tag> <suffix>      e.g.  summarize_prof.py r01f v6"""
import collections
import csv
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, suf = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    c = sqlite3.connect(os.path.join(src, "stats", "s_results.db"))
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(dst, "r01_bench_kernel_stats_%s.csv" % suf), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0].split("(")[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
    for r in rows[:12]:
        print("%-34s calls %4d  avg %8.1f us  min %8.1f  max %8.1f" % (r[0].split("(")[0][:34], r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "collect_traffic.py"),
                           os.path.join(src, "pmc_fetch", "f_counter_collection.csv"),
                           os.path.join(src, "pmc_write", "w_counter_collection.csv"), "16384",
                           os.path.join(dst, "traffic.json")])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, "pmc_sq", "q_counter_collection.csv"))):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        if not k.startswith(("k_agent", "k_coh", "k_field", "k_sp")):
            continue
        d = {cn: sum(x) / len(x) for cn, x in v.items()}
        d["valu_per_wave"] = d["SQ_INSTS_VALU"] / max(d["SQ_WAVES"], 1)
        # time the VALU instructions alone need at one wave64 instruction per 4 cycles per SIMD,
        # 1024 SIMDs, 2.4 GHz
        d["valu_issue_floor_us"] = d["SQ_INSTS_VALU"] / 1024 * 4 / 2.4e3
        out[k] = d
    doc = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES on bench.py "
                     "--steps 12 --warmup 3 (per-dispatch averages)",
           "valu_issue_floor": "SQ_INSTS_VALU / 1024 SIMDs x 4 cycles per wave64 instruction / 2.4 GHz",
           "kernels": out}
    json.dump(doc, open(os.path.join(dst, "r01_sq_counters_%s.json" % suf), "w"), indent=1)
    json.dump(doc, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1)      # read by bench.py
    for k in ("k_agent_step", "k_cohesion", "k_field_bfs<false>", "k_agent_pre"):
        if k in out:
            print(k, "VALU/wave %.0f  waves %.0f  issue floor %.1f us" % (out[k]["valu_per_wave"], out[k]["SQ_WAVES"], out[k]["valu_issue_floor_us"]))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "r01_bench_%s.json" % suf))
    shutil.copy(os.path.join(src, "fuzz.log"), os.path.join(dst, "r01_fuzz_gpu_%s.log" % suf))
    shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, "r01_pytest_gpu_%s.log" % suf))


if __name__ == "__main__":
    main()
