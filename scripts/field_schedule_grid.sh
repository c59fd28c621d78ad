cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_g
for i in 1 2; do
for after in neighbours start; do for cus in 128 160 192 256; do
  env NAVTICK_FIELDS_AFTER=$after NAVTICK_FIELD_CUS=$cus python bench.py --no-cpu-baseline --no-crowded --steps 20 --no-dropin 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('summary') or d; print('after=$after cus=$cus', round(d['ms_per_step'],4), round(s['ms_per_step_median'],4), s['ms_tick_5_50_100'])"
done; done; done > gpurun_out/r05_g/field_schedule_grid.txt 2>&1
cat gpurun_out/r05_g/field_schedule_grid.txt
