# DEVELOPER-ONLY: the band-pass kernel's workgroup order (PEBBLEGPU_FF_XCD=0: (block, channel) grid; default: XCD-aware) on one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -x -q -k "fastfir or config or two_stage or streambank" > gpurun_out/ab_ff_tests.log 2>&1 || { tail -20 gpurun_out/ab_ff_tests.log; exit 1; }
tail -1 gpurun_out/ab_ff_tests.log
for rep in 1 2; do
for x in 0 1; do
  for w in 2 3; do PEBBLEGPU_FF_XCD=$x python tools/ab_bank_pipe.py $w 400 2>&1 | tail -1 | sed "s/^/xcd=$x /"; done
  PEBBLEGPU_FF_XCD=$x python tools/bench_streambank.py 4 128 2>&1 | tail -2 | sed "s/^/xcd=$x /"
done
done
