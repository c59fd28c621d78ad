#!/bin/bash
# Developer session: ClearPath work counters (stats build) + SQ instruction counters of the crowded world.
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 400 python scripts/cp_stats.py --crowd > $OUT/cp_stats_crowd.json 2> $OUT/cp_stats.err; tail -c 200 $OUT/cp_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc -o p --output-format csv -- python bench.py --crowded --steps 8 --warmup 3 --no-cpu-baseline > $OUT/pmc.json 2> $OUT/pmc.err
tail -c 300 $OUT/pmc.err
python - <<'P' $OUT
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if "cp_" in k or "agent" in k:
        print(k, {c: (len(x), round(sum(x[-4:]) / len(x[-4:]) / 1e6, 2)) for c, x in v.items()})
P
