#!/bin/bash
# DEVELOPER-ONLY: in-run A/B of two builds of the library on the headline workload (same box, alternating, settled)
# usage: bash tools/ab_lib.sh OLD.so   (the tree's libpebblegpu.so is "new")
mkdir -p gpurun_out
cp pebblesdr_amd/libpebblegpu.so /tmp/lib_new.so
for rep in $(seq 1 ${REPS:-3}); do
  for v in old new; do
    if [ $v = old ]; then cp "$1" pebblesdr_amd/libpebblegpu.so; else cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so; fi
    python3 bench.py --steps 300 --warmup 10 --headline-only --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null || exit 1
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); r=d['roofline']; print('$v', 'step', d['ms_per_step'], 'spectrum co', r['avg_launch_ms'], 'alone', r.get('avg_launch_ms_alone'))"
  done
done
cp /tmp/lib_new.so pebblesdr_amd/libpebblegpu.so
