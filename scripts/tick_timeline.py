#!/usr/bin/env python3
"""The timeline of a few ticks out of a rocprofv3 --kernel-trace csv: for every kernel of tick T its start and end
relative to the first kernel of the tick (k_sp_count / k_sp_build), its queue, and the gaps on the chain.
    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ...
    python scripts/tick_timeline.py DIR [first_tick] [n_ticks]"""
import csv
import os
import sys


def find(d, suffix):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n.split("<")[0] + ("<" + n.split("<")[1][:6] if "<" in n else "")


def main():
    d = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    path = find(d, "kernel_trace.csv")
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the LAST world of the process whose ticks are back to back is hard to tell apart from the others: ticks are split
    # at every first-kernel-of-the-front; the main world's ticks are the first run of them
    marks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(("k_sp_count", "k_sp_build"))]
    if len(marks) < first + count + 1:
        print("only", len(marks), "ticks in the trace")
        return
    for t in range(first, first + count):
        lo, hi = marks[t], marks[t + 1]
        t0 = int(rows[lo]["Start_Timestamp"])
        # kernels that START inside the tick's window (the next tick's first kernel ends it), plus stragglers of the
        # previous tick still running
        print("---- tick %d: %d kernels, %.1f us to the next tick's first kernel" %
              (t, hi - lo, (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3))
        for r in rows[lo:hi]:
            s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            print("  %8.1f %8.1f  (%6.1f us)  q%-3s  %s" % (s, e, e - s, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))


if __name__ == "__main__":
    main()
