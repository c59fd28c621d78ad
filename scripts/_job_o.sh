cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash scripts/gpu_job.sh r06_o tests smoke bench bench20 cfgs hostov aux stats pmc fuzz
