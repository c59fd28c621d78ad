#!/bin/bash
# the binding: its GPU tests, then the drop-in timing of bench.py
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_binding_gpu.py -m gpu -x -q > $OUT/pytest_binding.log 2>&1; tail -4 $OUT/pytest_binding.log
timeout 600 python bench.py --steps 20 --no-crowded --no-sustained > $OUT/bench20.json 2> $OUT/bench20.err; python - <<P
import json
d = json.loads(open("$OUT/bench20.json").read().strip().splitlines()[-1])
print(d["ms_per_step"]); print(json.dumps(d.get("dropin"), indent=1))
P
