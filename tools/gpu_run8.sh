mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_t8.log 2>&1 || { tail -30 gpurun_out/r02_t8.log; exit 1; }
tail -3 gpurun_out/r02_t8.log
timeout -k 10 600 python3 bench.py > gpurun_out/r02_bench_u.json 2> gpurun_out/r02_bench_u.err || { tail -20 gpurun_out/r02_bench_u.err; exit 1; }
cat gpurun_out/r02_bench_u.json
