cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r02q; mkdir -p $OUT
gcc -shared -fPIC -o /tmp/segv.so scripts/_segv.c
LD_PRELOAD=/tmp/segv.so timeout 800 python -X faulthandler=0 -m pytest tests/test_agents_gpu.py tests/test_binding_gpu.py -q -s -p no:faulthandler > $OUT/segv.log 2>&1
echo EXIT $?
grep -n "SEGV" -A40 $OUT/segv.log | cut -c1-200 | head -60; tail -3 $OUT/segv.log
