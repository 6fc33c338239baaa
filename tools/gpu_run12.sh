mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_t12.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_t12.log
