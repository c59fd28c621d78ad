#!/bin/bash
# A/B only (no parity): `bash scripts/gpu_session3.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python scripts/ab_lib.py --run $@ --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -8 $OUT/ab_crowded.txt
timeout 900 python scripts/ab_lib.py --run $@ --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -8 $OUT/ab_100.txt
