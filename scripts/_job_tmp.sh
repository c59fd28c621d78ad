cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r02y; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-crowded > $OUT/bench.json 2> $OUT/prof.err
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/r02y/stats/s_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].split('(')[0]
    d[n].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for n in ("k_cp_rows","k_cp_heavy","k_agent_full",'k_agent_mid','k_agent_nbr','k_cohesion','void k_field_bfs<false>','k_sp_place'):
    v=[x[1]/1e3 for x in sorted(d[n])]
    print(n, len(v), ' '.join('%.0f'%x for x in v[::6]))
PY
