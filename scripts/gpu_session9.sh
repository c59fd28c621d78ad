#!/bin/bash
# crowded A/B + per-kernel durations of the crowded world per variant: `bash scripts/gpu_session9.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python scripts/ab_lib.py --run $@ --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -5 $OUT/ab_crowded.txt
for v in $@; do
  if [ $v != base ]; then export NAVHIP_LIB=$GRAFT_REPO_ROOT/build_prof/libnavhip_$v.so; else unset NAVHIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --crowded --warmup 3 --steps 10 > $OUT/bench_prof_$v.json 2>/dev/null)
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 $f | cut -c1-160 > $OUT/kstats_$v.csv
  echo "== $v"; cut -d, -f1-4 $OUT/kstats_$v.csv | cut -c1-110 | head -8
done
