#!/bin/bash
# One A/B session on the GPU box (gpurun): variants of libnavhip.so built beforehand with scripts/ab_lib.py --build /
# --build-rev (build_prof/libnavhip_NAME.so; `base` = the in-tree library), alternated
# inside ONE session because box-to-box noise (~3 %) is more than most single changes.
#
#   bash scripts/gpu_ab.sh <tag> [steps...] -- <variants...>
#
# steps (in the order given):
#   vparity   EVERY variant against the reference first (tests/test_agents_gpu.py + test_fields_gpu.py with NAVHIP_LIB = the
#             variant): a variant that is faster because it is wrong (COH_NP = 1, r05) must not reach the clock
#   parity    the ClearPath / velocity-step / whole-config parity tests (tests/test_agents_gpu.py, test_fullsize_ref_gpu.py)
#   binding   tests/test_binding_gpu.py + the drop-in timing of bench.py (--steps 20)
#   ab20      bench.py --steps 20 (the driver's window), 3 rounds
#   ab100     bench.py --steps 100, 2 rounds
#   crowded   bench.py --crowded --steps 20, 2 rounds
#   kstats20  per variant: rocprofv3 --kernel-trace --stats of the driver's window (20 ordinary ticks) -> kstats20_<variant>.csv
#   kstats    per variant: rocprofv3 --kernel-trace --stats of 10 crowded ticks -> kstats_<variant>.csv
#   sq        per variant: SQ / instruction-cache counters of 6 crowded ticks, three passes of four counters
#             (eight SQ counters in one pass crashed rocprofv3 on this stack) -> counters.txt
# Output: gpurun_out/<tag>/.  Example:  gpurun -- 'bash scripts/gpu_ab.sh r5a parity ab20 crowded -- base prev'
TAG=$1; shift
STEPS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do STEPS+=("$1"); shift; done
shift
VARS=${@:-base}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
pick() { if [ "$1" != base ]; then export NAVHIP_LIB=$GRAFT_REPO_ROOT/build_prof/libnavhip_${1}.so; else unset NAVHIP_LIB; fi; }
for s in "${STEPS[@]}"; do
case $s in
vparity) for v in $VARS; do pick $v
           timeout 300 python -m pytest tests/test_agents_gpu.py tests/test_fields_gpu.py -m gpu -x -q > $OUT/pytest_variant_$v.log 2>&1
           echo "variant $v: $(tail -1 $OUT/pytest_variant_$v.log)"
         done; unset NAVHIP_LIB ;;
parity)  timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_fullsize_ref_gpu.py -m gpu -x -q -k "clearpath or crowded or velocity_step or whole_config" > $OUT/pytest_cp.log 2>&1; tail -4 $OUT/pytest_cp.log ;;
binding) timeout 900 python -m pytest tests/test_binding_gpu.py -m gpu -x -q > $OUT/pytest_binding.log 2>&1; tail -3 $OUT/pytest_binding.log
         timeout 600 python bench.py --steps 20 --no-crowded --no-sustained > $OUT/bench20.json 2> $OUT/bench20.err
         python -c "import json,sys; d=json.loads(open('$OUT/bench20.json').read().strip().splitlines()[-1]); print(d['ms_per_step']); print(json.dumps(d.get('dropin'), indent=1))" ;;
ab20)    timeout 600 python scripts/ab_lib.py --run $VARS --steps=20 --rounds=3 > $OUT/ab_20.txt 2>&1; tail -$(( $(echo $VARS | wc -w) + 1 )) $OUT/ab_20.txt ;;
ab100)   timeout 900 python scripts/ab_lib.py --run $VARS --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -$(( $(echo $VARS | wc -w) + 1 )) $OUT/ab_100.txt ;;
crowded) timeout 600 python scripts/ab_lib.py --run $VARS --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -$(( $(echo $VARS | wc -w) + 1 )) $OUT/ab_crowded.txt ;;
kstats20) for v in $VARS; do pick $v
           (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof20_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crowded --no-sustained --steps 20 > $OUT/bench_prof20_$v.json 2>/dev/null)
           f=$(find /tmp/prof20_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f | cut -c1-160 > $OUT/kstats20_$v.csv
           echo "== $v"; cut -c1-110 $OUT/kstats20_$v.csv | head -12
         done; unset NAVHIP_LIB ;;
kstats)  for v in $VARS; do pick $v
           (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --crowded --warmup 3 --steps 10 > $OUT/bench_prof_$v.json 2>/dev/null)
           f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-160 > $OUT/kstats_$v.csv
           echo "== $v"; cut -c1-110 $OUT/kstats_$v.csv | head -8
         done; unset NAVHIP_LIB ;;
sq)      n=0
         for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
           n=$((n+1))
           for v in $VARS; do pick $v
             (cd /tmp && NAVHIP_HANDOVER=events timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${v}_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --crowded --warmup 3 --steps 6 > $OUT/pmc_${v}_$n.json 2>$OUT/pmc_${v}_$n.err)
             f=$(find /tmp/pmc_${v}_$n -name "*counter_collection.csv" | head -1)
             [ -n "$f" ] && python - "$f" "$v set$n" <<'P' | tee -a $OUT/counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:24]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in ("k_cp_heavy", "k_cp_rows", "k_agent_full", "k_cp_small"):
    for kk in acc:
        if kk.startswith(k):
            print(sys.argv[2], kk, {c: "%.4g" % v for c, v in sorted(acc[kk].items())})
P
           done
         done; unset NAVHIP_LIB ;;
esac
done
