# DEVELOPER-ONLY: A/B of the two-stage calls' hand-over (output buffers, event through the dispatch, where the host waits) on one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "two_stage or bench_geometry or config3 or config2 or interleaved" > gpurun_out/ab3_tests.log 2>&1 || { tail -20 gpurun_out/ab3_tests.log; exit 1; }
tail -2 gpurun_out/ab3_tests.log
for rep in 1 2; do
for cfg in "3 0 0" "3 1 0" "3 0 1" "3 1 1" "2 1 1"; do
  set -- $cfg
  for w in 2 3; do
    PEBBLEGPU_BANK_PIPE_BUFS=$1 PEBBLEGPU_BANK_PIPE_EXTEV=$2 PEBBLEGPU_BANK_PIPE_HOSTWAIT=$3 python tools/ab_bank_pipe.py $w 400 2>&1 | tail -1 | sed "s/^/bufs=$1 extev=$2 hostwait=$3 /"
  done
done
done
