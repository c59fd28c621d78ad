#!/usr/bin/env python3
"""Static loop census of one kernel in a hipcc --save-temps .s file (developer tool, no GPU needed):
every backward branch = a loop; prints its span, instruction counts by unit and the IR block name of its
header, so that the instruction cost of an inner loop can be read off without a profiler.

    python scripts/isa_loops.py file.s <mangled-name-substring> [min_len]
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
    end = next(i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end"))
    lab, ins, names = {}, [], {}
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(?:;\s*(.*))?", l)
        if m:
            lab[m.group(1)] = len(ins)
            names[m.group(1)] = (m.group(2) or "").strip()
            continue
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            ins.append(l.strip())
    print("%s: %d instructions" % (lines[start].split(":")[0][:70], len(ins)))
    loops = []
    for i, l in enumerate(ins):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in lab and lab[m.group(2)] <= i:
            loops.append((lab[m.group(2)], i, m.group(2)))
    loops.sort(key=lambda t: (t[0], -t[1]))
    for a, b, n in loops:
        if b - a + 1 < min_len:
            continue
        seg = ins[a:b + 1]
        depth = sum(1 for (a2, b2, _) in loops if a2 <= a and b2 >= b) - 1
        cnt = lambda p: sum(1 for x in seg if x.startswith(p))
        trans = sum(1 for x in seg if re.match(r"v_(sqrt|rsq|rcp|div_|exp|log)", x))
        print("%s%-10s [%6d..%6d] len %5d  valu %5d (trans/div %3d) salu %5d lds %4d vmem %3d  %s"
              % ("  " * depth, n, a, b, b - a + 1, cnt("v_"), trans, cnt("s_"), cnt("ds_"),
                 cnt("global_") + cnt("buffer_") + cnt("flat_") + cnt("scratch_"), names.get(n, "")[:60]))


if __name__ == "__main__":
    main()
