#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of two builds of the library on the bench's raw-int8 leg (and its float headline)
# usage: bash tools/ab_raw.sh OLD.so   (the tree's libpebblegpu.so is "new")
mkdir -p gpurun_out
cp pebblesdr_amd/libpebblegpu.so /tmp/lib_new.so
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp "$1" pebblesdr_amd/libpebblegpu.so; else cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so; fi
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$v', 'float2 step', d['ms_per_step'], 'int8 step', d['raw_int8']['ms_per_step'])"
  done
done
cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so
