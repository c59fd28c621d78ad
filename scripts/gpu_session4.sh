#!/bin/bash
# SQ counters of the crowded world for several library variants: `bash scripts/gpu_session4.sh <tag> <variants...>`
# (four counters per pass and a short timeout: eight SQ counters in one pass crashed rocprofv3 on this stack)
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in $@; do
  if [ $v != base ]; then export NAVHIP_LIB=$PWD/build_prof/libnavhip_$v.so; else unset NAVHIP_LIB; fi
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_$v -o p --output-format csv -- python bench.py --crowded --steps 8 --warmup 3 --no-cpu-baseline > $OUT/pmc_$v.json 2> $OUT/pmc_$v.err
  python - $OUT/pmc_$v $v <<'P'
import csv, sys, glob, collections
out, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in sorted(acc.items()):
    if "cp_heavy" in k:
        print(v, k, "us", round(sum(dur[k][-4:]) / 4), {n: round(sum(x[-4:]) / 4 / 1e6, 1) for n, x in c.items()})
P
done
