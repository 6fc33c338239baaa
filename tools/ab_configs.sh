#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of switches on the other BASELINE workloads (configs[2..4] legs of bench.py)
# usage: bash tools/ab_configs.sh "ENV=.." "ENV=.." ...
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    env $v python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/abc.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/abc.json')); print('$v', {k: (v['ms_per_step'], v['roofline']['kernel'], v['roofline'].get('avg_launch_ms')) for k, v in d.get('configs', {}).items()})"
  done
done
