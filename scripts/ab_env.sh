#!/bin/bash
# A/B two values of one environment knob on bench.py inside a single GPU session:
#   scripts/ab_env.sh NAVHIP_PRE_WG 64 256
var=$1; shift
for i in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms'].items()})"
  done
done
