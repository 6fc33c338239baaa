mkdir -p gpurun_out
PEBBLEGPU_T128_STAGGER=3 timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "spectrum or config2 or two_stream or full_size_properties" > gpurun_out/r02_t8.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02_t8.log
for st in 0 1 2 3 4 5 6 7; do
  echo "shift $st"; PEBBLEGPU_T128_STAGGER=$st python3 tools/bench_spectrum_sizes.py 8192 2>&1 | tail -1
done
