set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01l
python bench.py > gpurun_out/r01l/bench.json 2> gpurun_out/r01l/bench.err
tail -c 600 gpurun_out/r01l/bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r01l/stats -o s -- python bench.py > gpurun_out/r01l/bench_under_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r01l/pmc_fetch -o f --output-format csv -- python bench.py --steps 12 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r01l/pmc_write -o w --output-format csv -- python bench.py --steps 12 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/r01l/pmc_sq -o q --output-format csv -- python bench.py --steps 12 --warmup 3 > /dev/null 2>&1
find gpurun_out/r01l -name "*.csv" | head -20
python tests/tools/fuzz_gpu.py > gpurun_out/r01l/fuzz.log 2>&1; tail -1 gpurun_out/r01l/fuzz.log
python -m pytest tests -m gpu -q > gpurun_out/r01l/pytest_gpu.log 2>&1; tail -2 gpurun_out/r01l/pytest_gpu.log
