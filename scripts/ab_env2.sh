#!/bin/bash
# A/B combinations of environment knobs on bench.py inside ONE GPU session, alternating:  scripts/ab_env2.sh "A=1 B=hi" "A=0" ...
# (AB_ARGS: extra bench.py arguments; each variant = a quoted list of VAR=value)
for i in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-crowded ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('summary') or d; print('[$v]', round(d['ms_per_step'],4), round(s['ms_per_step_median'],4), s['ms_tick_5_50_100'], d['kernel_groups_ms_serial']['after_timed_region'])"
  done
done
