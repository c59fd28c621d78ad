#!/bin/bash
# A/B on the headline window (20 ticks) and 100 ticks: `bash scripts/gpu_session12.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python scripts/ab_lib.py --run $@ --steps=20 --rounds=3 > $OUT/ab_20.txt 2>&1; tail -4 $OUT/ab_20.txt
timeout 900 python scripts/ab_lib.py --run $@ --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -4 $OUT/ab_100.txt
