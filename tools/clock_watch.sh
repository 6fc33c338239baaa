#!/bin/bash
# DEVELOPER-ONLY: shader clock and socket power while the headline workload runs (is the spectrum kernel power-limited?)
# usage (GPU box): bash tools/clock_watch.sh [env assignments...]
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
rocm-smi --showclocks --showpower > gpurun_out/clock_idle.txt 2>&1
python3 bench.py --steps 6000 --warmup 10 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo
  sleep 0.7
done
wait $pid
python3 -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], r['avg_launch_ms'], r.get('avg_launch_ms_alone'))"
