mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "config3 or full_size_configs2 or chain_sweep or register_first or every_narrow or lifecycle or independent" > gpurun_out/r02_t5.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_t5.log
for L in 0 32 48 64; do echo "L=$L"; PEBBLEGPU_FUSED_L=$L python3 tools/bench_configs.py 2 2>&1 | tail -1; done
