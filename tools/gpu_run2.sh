set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "spectrum or config2 or config1 or two_stream or lifecycle or signal_strength or squelch" > gpurun_out/r02_t2.log 2>&1 || { tail -30 gpurun_out/r02_t2.log; exit 1; }
tail -3 gpurun_out/r02_t2.log
python3 bench.py --headline-only --no-cpu-baseline > gpurun_out/r02_b_q128.json 2> gpurun_out/r02_b_q128.err
cat gpurun_out/r02_b_q128.json
PEBBLEGPU_SPECTRUM_FREGS=1 python3 bench.py --headline-only --no-cpu-baseline > gpurun_out/r02_b_q128_fregs.json 2>&1
cat gpurun_out/r02_b_q128_fregs.json
PEBBLEGPU_SPECTRUM_SHARED=1 python3 bench.py --headline-only --no-cpu-baseline > gpurun_out/r02_b_shared.json 2>&1
cat gpurun_out/r02_b_shared.json
python3 tools/bench_spectrum_sizes.py > gpurun_out/r02_spec_sizes.txt 2>&1
cat gpurun_out/r02_spec_sizes.txt
