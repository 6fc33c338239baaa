mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "config3 or full_size_configs2 or chain_sweep or register_first or every_narrow or lifecycle or independent or bank" > gpurun_out/r02_t6.log 2>&1 || { tail -30 gpurun_out/r02_t6.log; exit 1; }
tail -3 gpurun_out/r02_t6.log
for L in 0 64 96 128 192 256; do echo "tp L=$L"; PEBBLEGPU_FUSED_L=$L timeout -k 10 120 python3 tools/bench_configs.py 2 2>&1 | tail -1; done
for L in 64; do echo "pipe L=$L"; PEBBLEGPU_FUSED_PIPE=1 PEBBLEGPU_FUSED_L=$L timeout -k 10 120 python3 tools/bench_configs.py 2 2>&1 | tail -1; done
