#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately, TCC has 4 slots)
of `bench.py` into profiles/traffic.json: HBM-side bytes per launch for the two phases of a tick.

Corrections, per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
  * both counters are in KB (x1024);
  * on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B: wide coalesced streaming reads are
    reported at exactly half -> doubled here ("fetch_corr"); other access widths are uncalibrated,
    so the raw figure is kept next to it;
  * WRITE_SIZE is calibrated on a known byte count in our own access pattern: k_field_bfs writes
    exactly 4096 B per request with 16 B/lane coalesced stores and nothing else.
Usage: collect_traffic.py <fetch_counter_csv> <write_counter_csv> <n_field_requests> <out_json>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fpath, wpath, nreq, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, nf = per_kernel(fpath, "FETCH_SIZE")
    write, nw = per_kernel(wpath, "WRITE_SIZE")
    fields_k = [k for k in fetch if k.startswith(("k_field_bfs", "k_field_generic"))]
    agent_k = [k for k in fetch if k.startswith(("k_sp_", "k_coh", "k_agent_"))]
    known_write = nreq * 4096.0
    bfs = [k for k in write if k.startswith("k_field_bfs")]
    wcal = known_write / write[bfs[0]] if bfs and write[bfs[0]] > 0 else None
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py",
        "per_kernel_fetch_bytes_raw": fetch, "per_kernel_write_bytes_raw": write,
        "dispatches_averaged": nf,
        "write_calibration": {"kernel": bfs[0] if bfs else None, "known_bytes": known_write,
                              "reported_bytes": write.get(bfs[0]) if bfs else None, "factor": wcal},
    }
    for name, ks in (("fields", fields_k), ("agents", agent_k)):
        f_raw = sum(fetch.get(k, 0.0) for k in ks)
        w_raw = sum(write.get(k, 0.0) for k in ks)
        res[name + "_fetch_bytes_raw"] = f_raw
        res[name + "_fetch_bytes_corr"] = 2.0 * f_raw
        res[name + "_write_bytes_raw"] = w_raw
        res[name + "_write_bytes_corr"] = w_raw * (wcal if wcal else 1.0)
        res[name + "_bytes_per_launch"] = 2.0 * f_raw + w_raw * (wcal if wcal else 1.0)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k.endswith("_per_launch") or k == "write_calibration"}))


if __name__ == "__main__":
    main()
