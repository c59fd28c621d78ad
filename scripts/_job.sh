cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_regions_gpu.py tests/test_multirank_gpu.py tests/test_agents_gpu.py tests/test_edge_gpu.py tests/test_fullsize_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
timeout 300 python scripts/rank_cost_probe.py 1 4 8 2>&1 | tail -3
timeout 300 python scripts/rank_cost_probe.py 1 4 8 2>&1 | tail -3
