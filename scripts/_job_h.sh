cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r06_h; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_golden_gpu.py tests/test_agents_gpu.py tests/test_tick_gpu.py tests/test_pool_gpu.py tests/test_edge_gpu.py tests/test_fullsize_ref_gpu.py -m gpu -x -q > $OUT/pytest_step.log 2>&1; tail -6 $OUT/pytest_step.log
bash scripts/gpu_job.sh r06_h bench20
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-crowded --no-sustained --no-dropin --no-weak > $GRAFT_REPO_ROOT/$OUT/bench_traced.json 2>/dev/null)
python scripts/tick_timeline.py /tmp/tl 12 2 > $OUT/timeline_cfg2.txt 2>&1; head -60 $OUT/timeline_cfg2.txt
for cfg in 0 2of8; do timeout 300 python scripts/queue_probe.py --config $cfg --reps 3 --ticks 40 2>&1 | grep -E "rep|spread"; done | tee $OUT/queue_small.txt
