#!/bin/bash
# One A/B + parity session on the GPU box: `bash scripts/gpu_session.sh <tag> <variants...>` (variants built
# beforehand with scripts/ab_lib.py --build*, `base` = the in-tree library).
TAG=$1; shift
VARS=${@:-base prev}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_fullsize_ref_gpu.py -m gpu -x -q -k "clearpath or crowded or velocity_step" > $OUT/pytest_cp.log 2>&1; tail -5 $OUT/pytest_cp.log
timeout 900 python scripts/ab_lib.py --run $VARS --crowded --steps=20 --rounds=2 > $OUT/ab_crowded.txt 2>&1; tail -8 $OUT/ab_crowded.txt
timeout 900 python scripts/ab_lib.py --run $VARS --steps=100 --rounds=2 > $OUT/ab_100.txt 2>&1; tail -8 $OUT/ab_100.txt
if [ -f build_prof/libnavhip_cpstats.so ]; then
  timeout 400 python scripts/cp_stats.py --crowd > $OUT/cp_stats_crowd.json 2> $OUT/cp_stats.err; tail -c 300 $OUT/cp_stats.err
  timeout 400 python scripts/cp_stats.py > $OUT/cp_stats.json 2>> $OUT/cp_stats.err
fi
