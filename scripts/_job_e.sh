cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r06_e; mkdir -p $OUT
NAVHIP_STREAM_DEBUG=1 timeout 600 python scripts/queue_probe.py --config 2of8 --reps 10 --ticks 40 > $OUT/queue_pipes_2of8.txt 2>&1; grep -E "rep|spread|same pipe" $OUT/queue_pipes_2of8.txt | head -40
for cfg in 0 2; do
  timeout 600 python scripts/queue_probe.py --config $cfg --reps 10 --ticks 40 > $OUT/queue_pipes_$cfg.txt 2>&1; grep -E "rep|spread" $OUT/queue_pipes_$cfg.txt
done
timeout 400 python scripts/queue_probe.py --config 2of8 --reps 10 --driver python 2>&1 | grep -E "rep|spread" | tee $OUT/queue_pipes_python_2of8.txt
timeout 900 python -m pytest tests/test_tick_gpu.py -m gpu -x -q > $OUT/pytest_tick.log 2>&1; tail -5 $OUT/pytest_tick.log
bash scripts/gpu_job.sh r06_e bench20
