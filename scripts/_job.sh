cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3w
python scripts/ab_lib.py --run base prev --rounds=3 --steps=100 > gpurun_out/r3w/ab100.txt 2>&1; tail -3 gpurun_out/r3w/ab100.txt
python scripts/ab_lib.py --run base prev --rounds=2 --steps=20 --crowded > gpurun_out/r3w/abcrowd.txt 2>&1; tail -3 gpurun_out/r3w/abcrowd.txt
timeout 900 python -m pytest tests/test_agents_gpu.py tests/test_fullsize_ref_gpu.py tests/test_blockers_gpu.py tests/test_binding_gpu.py -m gpu -x -q 2>&1 | grep -v -i "rccl\|hip version\|rocm version\|hostname" | tail -4
bash scripts/gpu_job.sh r3w pmc > /dev/null 2>&1
