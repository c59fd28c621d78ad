cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash scripts/gpu_job.sh r06_l tests smoke bench20 stats pmc
