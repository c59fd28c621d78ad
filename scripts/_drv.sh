cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py "$@" --no-cpu-baseline --no-crowded 2>/tmp/e.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), [x and round(x,3) for x in d['ms_tick_5_50_100']])"; tail -2 /tmp/e.txt | grep -v amdgpu; }
for i in 1 2 3; do run --gpus 1 --steps 20 --warmup 5; done
for i in 1 2; do run; done
