# DEVELOPER-ONLY: the headline's first-stage edges in the lean kernel's own launch (default) against a launch of k_mix_hb11_bank behind it
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/ab_le_tests.log 2>&1 || { tail -30 gpurun_out/ab_le_tests.log; exit 1; }
tail -1 gpurun_out/ab_le_tests.log
for rep in 1 2 3; do
for e in 1 0; do
  PEBBLEGPU_LEAN_EDGE_LAUNCH=$e python bench.py --no-cpu-baseline --headline-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('edge launch=$e', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['raw_int8'] if 'raw_int8' in d else '')"
done
done
