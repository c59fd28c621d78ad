cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2000 python scripts/ab_lib.py --run r2 base c046 --rounds=2 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -10
