#!/bin/bash
# W7: is the bimodal tick time the sharing of hardware queues between streams?  (scripts/queue_probe.py)
#   gpurun -- 'bash scripts/gpu_queue_probe.sh <tag>'
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  for cfg in 2of8 0; do
    env "$@" timeout 400 python scripts/queue_probe.py --config $cfg --reps 6 --ticks 40 > $OUT/queue_${name}_$cfg.txt 2>&1
    grep -E "rep|spread" $OUT/queue_${name}_$cfg.txt
  done
}
echo "== pooled (as shipped)";            run pooled X=1
echo "== aux0 dedicated";                 run ded1 NAVHIP_AUX_DEDICATED=1
echo "== aux0 + aux1 dedicated";          run ded3 NAVHIP_AUX_DEDICATED=3
echo "== pooled, GPU_MAX_HW_QUEUES=8";    run hwq8 GPU_MAX_HW_QUEUES=8
echo "== python driver, pooled";          env X=1 timeout 400 python scripts/queue_probe.py --config 2of8 --reps 6 --driver python 2>&1 | grep -E "rep|spread" | tee $OUT/queue_pooled_python_2of8.txt
echo "== python driver, dedicated";       env NAVHIP_AUX_DEDICATED=3 timeout 400 python scripts/queue_probe.py --config 2of8 --reps 6 --driver python 2>&1 | grep -E "rep|spread" | tee $OUT/queue_ded3_python_2of8.txt
