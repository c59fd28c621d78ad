#!/bin/bash
# A/B two values of one environment knob on bench.py inside a single GPU session:
#   scripts/ab_env.sh NAVTICK_FIELD_CUS 160 192 224
var=$1; shift
for i in 1 2 3; do
  for v in "$@"; do
    env $var=$v python bench.py --no-cpu-baseline --no-crowded ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('summary') or d; print('$var=$v', round(d['ms_per_step'],4), round(s['ms_per_step_median'],4), s['ms_tick_5_50_100'], d['config'].get('tick_driver'))"
  done
done
