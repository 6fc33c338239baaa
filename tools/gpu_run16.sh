mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_t16.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_t16.log
python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/r02_b16.json 2>gpurun_out/r02_b16.err; python3 -c "
import json;d=json.load(open('gpurun_out/r02_b16.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_alone']); print({k:(v['ms_per_step']) for k,v in d['configs'].items()}); print(d['raw_int8'])"
