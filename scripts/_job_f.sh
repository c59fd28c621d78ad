cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r06_f; mkdir -p $OUT
for cfg in 2of8 0 2; do
  NAVHIP_STREAM_DEBUG=1 timeout 600 python scripts/queue_probe.py --config $cfg --reps 10 --ticks 40 > $OUT/queue_own_$cfg.txt 2>&1; grep -E "rep|spread" $OUT/queue_own_$cfg.txt
done
timeout 400 python scripts/queue_probe.py --config 2of8 --reps 10 --driver python 2>&1 | grep -E "rep|spread" | tee $OUT/queue_own_python_2of8.txt
timeout 900 python -m pytest tests/test_tick_gpu.py -m gpu -x -q > $OUT/pytest_tick.log 2>&1; tail -5 $OUT/pytest_tick.log
bash scripts/gpu_job.sh r06_f bench20
