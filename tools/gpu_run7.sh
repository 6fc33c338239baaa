mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_t7.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_t7.log
PEBBLEGPU_SPECTRUM_SHARED=1 timeout -k 10 600 python3 bench.py > gpurun_out/r02_bench7.json 2> gpurun_out/r02_bench7.err; echo "bench rc=$?"; tail -3 gpurun_out/r02_bench7.err; cat gpurun_out/r02_bench7.json
