#!/bin/bash
# VERDICT r04 item 5, sized before it is built: what would stepping agents in cell order save AT BEST?  NAVTICK_UID_ORDER
# numbers the entities of the bench world in spatial order (tick.py), so that the uid-order kernels are cell-order kernels
# with every per-entity array still contiguous.  Per variant: the tick (3 rounds) and the per-kernel durations of 20 ticks.
#   bash scripts/uid_order_probe.sh <tag>
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for r in 1 2 3; do
  for v in none cell flock_cell; do
    if [ $v = none ]; then unset NAVTICK_UID_ORDER; else export NAVTICK_UID_ORDER=$v; fi
    timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-crowded --no-sustained 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('uid_order=$v', round(d['ms_per_step'],4), d['kernel_groups_ms_serial']['after_timed_region'])"
  done
done > $OUT/uid_order.txt 2>&1
for v in none cell flock_cell; do
  if [ $v = none ]; then unset NAVTICK_UID_ORDER; else export NAVTICK_UID_ORDER=$v; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/uo_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-crowded --no-sustained > /dev/null 2>&1)
  f=$(find /tmp/uo_$v -name "*kernel_stats.csv" | head -1)
  echo "== uid_order=$v" >> $OUT/uid_order.txt
  [ -n "$f" ] && python - "$f" >> $OUT/uid_order.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_agent", "k_cp_", "k_cohesion", "k_sp_place", "k_field_bfs")):
        print("%-22s calls %4s avg %8.1f us" % (n[:22], r["Calls"], float(r["AverageNs"]) / 1e3))
P
done
for v in none flock_cell; do
  if [ $v = none ]; then unset NAVTICK_UID_ORDER; else export NAVTICK_UID_ORDER=$v; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/uof_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-crowded --no-sustained > /dev/null 2>&1)
  f=$(find /tmp/uof_$v -name "*counter_collection.csv" | head -1)
  echo "== FETCH_SIZE (KB as counted, x2 on gfx950) per launch, uid_order=$v" >> $OUT/uid_order.txt
  [ -n "$f" ] && python - "$f" >> $OUT/uid_order.txt <<'P'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:22]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    if k.startswith(("k_agent", "k_cp_", "k_cohesion", "k_sp_", "k_coh")):
        print("%-22s launches %4d  mean %10.0f KB" % (k, len(v), sum(v) / len(v)))
P
done
cat $OUT/uid_order.txt
