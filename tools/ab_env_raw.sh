#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of environment switches on the bench's float2 headline and its raw-int8 leg
# usage: bash tools/ab_env_raw.sh "ENV=.." "ENV=.." ...
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    env $v python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$v', 'float2 step', d['ms_per_step'], 'int8 step', d['raw_int8']['ms_per_step'])"
  done
done
