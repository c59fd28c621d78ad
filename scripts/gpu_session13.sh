#!/bin/bash
# ClearPath parity + A/B (20 ticks, 100 ticks, crowded): `bash scripts/gpu_session13.sh <tag> <variants...>`
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_fullsize_ref_gpu.py -m gpu -x -q -k "clearpath or crowded or velocity_step or whole_config" > $OUT/pytest_cp.log 2>&1; tail -4 $OUT/pytest_cp.log
timeout 600 python scripts/ab_lib.py --run $@ --steps=20 --rounds=3 > $OUT/ab_20.txt 2>&1; tail -3 $OUT/ab_20.txt
timeout 900 python scripts/ab_lib.py --run $@ --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -3 $OUT/ab_100.txt
timeout 600 python scripts/ab_lib.py --run $@ --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -3 $OUT/ab_crowded.txt
