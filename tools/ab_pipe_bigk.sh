cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "two_stage or bench_geometry or config3 or config2 or interleaved or sweep" > gpurun_out/ab3_tests.log 2>&1 || { tail -20 gpurun_out/ab3_tests.log; exit 1; }
tail -1 gpurun_out/ab3_tests.log
for k in 8 16 32 128; do c=$((3200/k)); [ $c -lt 30 ] && c=30; python tools/ab_bank_pipe.py 2 $c $k | tail -1; done
python tools/ab_bank_pipe.py 3 400 | tail -1
