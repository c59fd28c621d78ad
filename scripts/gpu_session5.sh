#!/bin/bash
# parity of the round's new paths + the driver's bench invocation
TAG=$1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_fullsize_ref_gpu.py tests/test_agents_gpu.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; tail -12 $OUT/pytest_new.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; tail -c 1500 $OUT/bench_steps20.json; tail -5 $OUT/bench_steps20.err
