cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r06_g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_tick_gpu.py -m gpu -x -q > $OUT/pytest_tick.log 2>&1; tail -5 $OUT/pytest_tick.log
NAVTICK_TORCH_STREAM=1 NAVHIP_STREAM_DEBUG=1 timeout 600 python scripts/queue_probe.py --config 2of8 --reps 12 --ticks 40 > $OUT/queue_torch_2of8.txt 2>&1; grep -E "rep|spread" $OUT/queue_torch_2of8.txt
bash scripts/gpu_job.sh r06_g tests smoke
