#!/bin/bash
# instruction-cache / wait counters of the crowded world per variant: `bash scripts/gpu_session10.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
MODE=${CP_BENCH_MODE:---crowded}
n=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQC_ICACHE_MISSES_DUPLICATE"; do
  n=$((n+1))
  for v in $@; do
    if [ $v != base ]; then export NAVHIP_LIB=$GRAFT_REPO_ROOT/build_prof/libnavhip_$v.so; else unset NAVHIP_LIB; fi
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${v}_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $MODE --warmup 3 --steps 6 > $OUT/pmc_${v}_$n.json 2>$OUT/pmc_${v}_$n.err)
    f=$(find /tmp/pmc_${v}_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$v set$n" <<'P' | tee -a $OUT/counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:24]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in ("k_cp_heavy", "k_cp_rows", "k_agent_full", "k_cp_small"):
    for kk in acc:
        if kk.startswith(k):
            print(sys.argv[2], kk, {c: "%.4g" % v for c, v in sorted(acc[kk].items())})
P
  done
done
