#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of switches on the headline workload (same box, alternating, settled)
# usage: bash tools/ab_spectrum.sh "ENV=.." "ENV=.. ENV2=.." ...
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    env $v python3 bench.py --steps 300 --warmup 10 --headline-only --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$v', 'step', d['ms_per_step'], 'spectrum co', r['avg_launch_ms'], 'alone', r.get('avg_launch_ms_alone'))"
  done
done
