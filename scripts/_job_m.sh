cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r06_m; mkdir -p $OUT
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-crowded --no-sustained > $OUT/b$i.json 2>$OUT/b$i.err; python -c "
import json; d=json.loads(open('$OUT/b$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['summary']['host_enqueue_ms'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['roofline']['hbm_frac_measured'])"; done
nproc; cat /proc/loadavg
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tl -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-crowded --no-sustained --no-dropin > /dev/null 2>$GRAFT_REPO_ROOT/$OUT/prof.err; echo "rocprof exit $?")
tail -3 $OUT/prof.err
