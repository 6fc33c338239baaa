mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_t4.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_t4.log
python3 tools/bench_configs.py 2 > gpurun_out/r02_c2_fused.json 2>&1; cat gpurun_out/r02_c2_fused.json
for L in 32 64 128; do echo "L=$L"; PEBBLEGPU_FUSED_L=$L python3 tools/bench_configs.py 2 2>&1 | tail -1; done
PEBBLEGPU_NO_FUSED_DEC=1 python3 tools/bench_configs.py 2 2>&1 | tail -1
