#!/bin/bash
# A/B only (crowded, 100 ticks): `bash scripts/gpu_session8.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python scripts/ab_lib.py --run $@ --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -7 $OUT/ab_crowded.txt
timeout 900 python scripts/ab_lib.py --run $@ --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -7 $OUT/ab_100.txt
