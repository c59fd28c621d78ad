#!/usr/bin/env python3
"""Turn the outputs of `scripts/gpu_job.sh <tag> tests bench calib stats pmc fuzz aux` (gpurun_out/<tag>/)
into the committed summaries under profiles/: kernel stats csv, traffic.json (PMC FETCH/WRITE passes),
SQ counter summary, VALU calibration, bench line, logs.  Everything measured is stamped with the sha of
the kernel sources it was measured on (bench.csrc_sha): bench.py refuses to quote traffic / counters
of another tree.
Usage: summarize_prof.py <tag> <suffix>      e.g.  summarize_prof.py r02z a"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench    # noqa: E402

EARLY = (26, 32)  # the ticks bench.py profiles behind the DRIVER's invocation (--steps 20 --warmup 5): counters and
                  # durations of one line come from the same window of the world
LAST = 6        # ticks that count: the serial, profiled ticks at the END of bench.py (the
                # ticks its roofline times) -- not the average over a run in which the world crowds


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(path):
    """{kernel: {counter: [values in dispatch order]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def find(src, sub, name):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(name):
                return os.path.join(root, f)
    return None


def main():
    tag, suf = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    sha = bench.csrc_sha()
    pre = sys.argv[3] if len(sys.argv) > 3 else "r06_"

    def copy(name, out):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, out))
            return True
        print("missing", name)
        return False

    # ---- rocprofv3 --kernel-trace --stats of bench.py ---------------------------------------------
    ks = find(src, "stats", "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(dst, pre + "bench_kernel_stats_%s.csv" % suf))
        for r in list(csv.DictReader(open(ks)))[:14]:
            print("%-30s calls %5s  avg %9.1f us  %5s %%" % (short(r["Name"])[:30], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    ks20 = find(src, "stats20", "kernel_stats.csv")
    if ks20:      # the driver's own invocation (--steps 20 --warmup 5) under the profiler
        shutil.copy(ks20, os.path.join(dst, pre + "bench_steps20_kernel_stats_%s.csv" % suf))
        copy("bench_under_prof20.json", pre + "bench_steps20_under_prof_%s.json" % suf)
    tr = find(src, "stats", "kernel_trace.csv")
    if tr:
        # per-kernel duration at ticks 5 / 50 / 100 of the traced run (the world crowds: one average hides it)
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(tr)):
            d[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        per = {}
        for k, v in d.items():
            v.sort()
            if len(v) >= 100:
                per[k] = {"launches": len(v), "us_at_tick_5_50_100": [round(v[i][1], 1) for i in (9, 54, min(104, len(v) - 1))],
                          "us_mean": round(sum(x[1] for x in v) / len(v), 1)}
        json.dump({"source": "rocprofv3 --kernel-trace of bench.py --steps 100 --warmup 5 (gpurun_out/%s)" % tag,
                   "csrc_sha": sha, "kernels": per}, open(os.path.join(dst, pre + "kernel_durations_%s.json" % suf), "w"), indent=1)

    # ---- PMC passes ------------------------------------------------------------------------------
    fpath = find(src, "pmc_FETCH_SIZE", "counter_collection.csv")
    wpath = find(src, "pmc_WRITE_SIZE", "counter_collection.csv")
    if fpath and wpath:
        F, Wr = counters(fpath), counters(wpath)

        def per_tick(acc, k):
            """launches of kernel k per tick (k_cp_rows and the scans are launched twice a tick)"""
            ref = max((len(x.get(c, [])) for kk, x in acc.items() if kk.startswith("k_agent_mid") for c in x), default=0)
            n = max((len(x) for x in acc.get(k, {}).values()), default=0)
            return max(1, round(n / ref)) if ref else 1

        def last(acc, k, c):
            """bytes per TICK: the mean over the last LAST ticks' dispatches, times the launches per tick"""
            m = per_tick(acc, k)
            v = acc.get(k, {}).get(c, [])
            v = v[-LAST * m:]
            return sum(v) / len(v) * m * 1024.0 if v else 0.0

        def early(acc, k, c):
            """the same for the ticks EARLY of the run"""
            m = per_tick(acc, k)
            v = acc.get(k, {}).get(c, [])
            v = v[EARLY[0] * m:EARLY[1] * m]
            return sum(v) / len(v) * m * 1024.0 if v else 0.0
        fetch = {k: last(F, k, "FETCH_SIZE") for k in F}
        write = {k: last(Wr, k, "WRITE_SIZE") for k in Wr}
        fetch_e = {k: early(F, k, "FETCH_SIZE") for k in F}
        write_e = {k: early(Wr, k, "WRITE_SIZE") for k in Wr}
        bfs = [k for k in write if k.startswith("k_field_bfs")]
        known = 16384 * 4096.0
        wcal = known / write[bfs[0]] if bfs and write[bfs[0]] > 0 else None
        res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py "
                         "--steps 100 --warmup 5; per kernel and TICK the mean over the last %d ticks (the profiled ticks "
                         "at the end of the run, which bench.py's roofline times)" % LAST,
               "early_window": "ticks %d-%d" % (EARLY[0] + 1, EARLY[1]),
               "files": bench.csrc_files(),
               "corrections": "both counters are KB (x1024); gfx950 FETCH_SIZE counts 128-B requests at 64 B: x2 "
                              "('corr'); WRITE_SIZE calibrated on k_field_bfs, which writes exactly 4096 B per field "
                              "with 16 B/lane coalesced stores",
               "per_kernel_fetch_bytes_raw": fetch, "per_kernel_write_bytes_raw": write,
               "write_calibration": {"kernel": bfs[0] if bfs else None, "known_bytes": known,
                                     "reported_bytes": write.get(bfs[0]) if bfs else None, "factor": wcal}}
        groups = {"fields": ("k_field_bfs", "k_field_generic"), "agents": ("k_sp_", "k_coh", "k_agent_", "k_cp_", "k_wl_", "k_zero")}
        for name, pfx in groups.items():
            f_raw = sum(v for k, v in fetch.items() if k.startswith(pfx))
            w_raw = sum(v for k, v in write.items() if k.startswith(pfx))
            res[name + "_fetch_bytes_raw"] = f_raw
            res[name + "_fetch_bytes_corr"] = 2.0 * f_raw
            res[name + "_write_bytes_corr"] = w_raw * (wcal or 1.0)
            res[name + "_bytes_per_launch"] = 2.0 * f_raw + w_raw * (wcal or 1.0)
            res[name + "_bytes_per_launch_early"] = 2.0 * sum(v for k, v in fetch_e.items() if k.startswith(pfx)) \
                + (wcal or 1.0) * sum(v for k, v in write_e.items() if k.startswith(pfx))
        res["covers"] = bench.stamp_units(res["files"], bench.stamp_kernels(res))
        res["csrc_sha"], res["csrc_sha_all_files"] = bench.csrc_sha(files=res["covers"]), sha
        json.dump(res, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
        json.dump(res, open(os.path.join(dst, pre + "traffic_%s.json" % suf), "w"), indent=1)
        print({k: v for k, v in res.items() if k.endswith("_per_launch")}, "write calibration", wcal)
    qpath = find(src, "pmc_SQ_INSTS_VALU", "counter_collection.csv")
    if qpath:
        Q = counters(qpath)
        out = {}
        for k, v in Q.items():
            if not k.startswith(("k_agent", "k_coh", "k_field", "k_sp", "k_cp", "k_wl", "k_zero")):
                continue
            ref = max((len(x) for kk, vv in Q.items() if kk.startswith("k_agent_mid") for x in vv.values()), default=0)
            n = max((len(x) for x in v.values()), default=0)
            m = max(1, round(n / ref)) if ref else 1                 # launches per tick
            dd = {cn: sum(x[-LAST * m:]) / len(x[-LAST * m:]) * m for cn, x in v.items()}
            dd["launches_per_tick"] = m
            dd["valu_per_wave"] = dd.get("SQ_INSTS_VALU", 0) / max(dd.get("SQ_WAVES", 1), 1)
            early = {cn: sum(x[EARLY[0] * m:EARLY[1] * m]) / max(1, len(x[EARLY[0] * m:EARLY[1] * m])) * m for cn, x in v.items()}
            dd["SQ_INSTS_VALU_early_ticks"] = early.get("SQ_INSTS_VALU")
            out[k] = dd
        doc = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES on "
                         "bench.py --steps 100 --warmup 5; per kernel and TICK (a kernel launched twice a tick counts "
                         "twice): the mean over the last %d ticks (early ticks: %d-%d)" % (LAST, EARLY[0] + 1, EARLY[1]), "files": bench.csrc_files(), "kernels": out}
        doc["covers"] = bench.stamp_units(doc["files"], bench.stamp_kernels(doc))
        doc["csrc_sha"], doc["csrc_sha_all_files"] = bench.csrc_sha(files=doc["covers"]), sha
        json.dump(doc, open(os.path.join(dst, "sq_counters.json"), "w"), indent=1)
        json.dump(doc, open(os.path.join(dst, pre + "sq_counters_%s.json" % suf), "w"), indent=1)
        for k in sorted(out, key=lambda k: -out[k].get("SQ_INSTS_VALU", 0))[:8]:
            print("%-24s VALU wave-instr/launch %12.0f (early %12.0f)  waves %8.0f" % (
                k, out[k].get("SQ_INSTS_VALU", 0), out[k].get("SQ_INSTS_VALU_early_ticks") or 0, out[k].get("SQ_WAVES", 0)))
    if os.path.exists(os.path.join(src, "valu_calib.json")):
        try:
            c = json.load(open(os.path.join(src, "valu_calib.json")))
            json.dump(c, open(os.path.join(dst, pre + "valu_calib.json"), "w"), indent=1)
        except Exception as e:
            print("valu_calib.json:", e)
    copy("bench.json", pre + "bench_%s.json" % suf)
    copy("bench_steps20.json", pre + "bench_%s_steps20.json" % suf)
    for c in (0, 1, 3, 4):
        copy("bench_cfg%d.json" % c, pre + "bench_cfg%d_%s.json" % (c, suf))
    copy("bench_secondary.json", pre + "bench_secondary_%s.json" % suf)
    sk = find(src, "secondary", "kernel_stats.csv")
    if sk:
        shutil.copy(sk, os.path.join(dst, pre + "secondary_kernel_stats_%s.csv" % suf))
    copy("pytest_gpu.log", pre + "pytest_gpu_%s.log" % suf)
    copy("fuzz.log", pre + "fuzz_gpu_%s.log" % suf)
    copy("bench_aux.json", pre + "bench_aux_%s.json" % suf)
    copy("cp_unit_hist.json", pre + "cp_unit_hist_%s.json" % suf)


if __name__ == "__main__":
    main()
