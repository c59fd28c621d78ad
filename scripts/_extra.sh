cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r3J; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
NAVHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; tail -c 400 $OUT/bench_2ranks_gloo.json; tail -2 $OUT/bench_2ranks_gloo.err
