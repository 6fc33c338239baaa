mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --deselect "tests/test_parity_gpu.py::test_tune_only_mode_freezes_what_lies_behind_it" > gpurun_out/r02_t11.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_t11.log
python3 tools/bench_streambank.py 4 128 2>&1 | tail -1
python3 tools/diag/agc_am.py 2>&1 | tail -24
