cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 2000 python scripts/ab_lib.py --run base nbr8 occ3 occ2 --rounds=2 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12
