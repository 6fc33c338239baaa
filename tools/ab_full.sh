#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of two builds of the library on the whole bench line (headline, raw-int8 leg, the other configs)
# usage: bash tools/ab_full.sh OLD.so   (the tree's libpebblegpu.so is "new")
mkdir -p gpurun_out
cp pebblesdr_amd/libpebblegpu.so /tmp/lib_new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp "$1" pebblesdr_amd/libpebblegpu.so; else cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so; fi
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$v', 'float2', d['ms_per_step'], 'int8', d['raw_int8']['ms_per_step'], {k: v['ms_per_step'] for k, v in d['configs'].items()})"
  done
done
cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so
