#!/bin/bash
# parity of the ClearPath paths + A/B over 100 ticks (sustained regime) and the crowded world
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_fullsize_ref_gpu.py tests/test_multirank_gpu.py -m gpu -x -q -k "clearpath or crowded or velocity_step or whole_config or shared or split_one_world" > $OUT/pytest_cp.log 2>&1; tail -5 $OUT/pytest_cp.log
timeout 900 python scripts/ab_lib.py --run $@ --steps=100 --rounds=3 > $OUT/ab_100.txt 2>&1; tail -8 $OUT/ab_100.txt
timeout 600 python scripts/ab_lib.py --run $@ --crowded --steps=20 --rounds=1 > $OUT/ab_crowded.txt 2>&1; tail -4 $OUT/ab_crowded.txt
timeout 300 python scripts/rank_cost_probe.py --strong 1 2 4 8 > $OUT/rank_cost_strong.txt 2>&1; tail -5 $OUT/rank_cost_strong.txt
