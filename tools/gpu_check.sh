#!/bin/bash
# the round's routine GPU check: the GPU test suite, then the bench line (stderr kept)
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 600 python3 bench.py "$@" > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err || { tail -20 gpurun_out/bench_check.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_check.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r.get('avg_launch_ms_alone'), r.get('measured_copy_peak_GBs')); print({k: v['ms_per_step'] for k, v in d.get('configs', {}).items()}, d.get('raw_int8', {}).get('ms_per_step'))"
