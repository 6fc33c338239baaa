#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call:  gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# (rocprofv3 always gets the interpreter itself after `--`; counter passes carry --kernel-trace only.)
tag=${1:-rXX}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
set -e
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $o/${tag}_gpu_tests.log 2>&1 || { tail -30 $o/${tag}_gpu_tests.log; exit 1; }
tail -2 $o/${tag}_gpu_tests.log
timeout -k 10 600 python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err
# per-kernel durations of the bench command
rm -rf $o/${tag}_trace; rocprofv3 --kernel-trace --stats -d $o/${tag}_trace -- python3 bench.py --steps 20 --no-cpu-baseline > $o/${tag}_trace.log 2>&1
python3 tools/kernel_stats.py $o/${tag}_trace $o/${tag}_kernel_stats.csv > /dev/null
# HBM traffic of the headline workload, the configs[2] bank, the configs[3] shard and the configs[4] shard (separate FETCH_SIZE / WRITE_SIZE passes)
export PMC_CALIBRATE=1
for w in head configs2 configs3 configs4; do
  case $w in head) script="tools/pmc_workload.py";; configs2) script="tools/pmc_bank.py 2";; configs3) script="tools/pmc_bank.py 3";; configs4) script="tools/pmc_streambank.py";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $o/${tag}_pmc_${w}_$c; rocprofv3 --pmc $c --kernel-trace -d $o/${tag}_pmc_${w}_$c -- python3 $script > $o/${tag}_pmc_${w}_$c.log 2>&1
  done
  python3 tools/parse_traffic.py $o/${tag}_pmc_${w}_FETCH_SIZE $o/${tag}_pmc_${w}_WRITE_SIZE $o/${tag}_traffic_${w}.json > /dev/null
done
unset PMC_CALIBRATE
# SQ counters of the kernels the round worked on
bash tools/run_sq_counters.sh ${tag}_fused_dec k_mix_dec_mfma tools/pmc_bank.py 2 > /dev/null
bash tools/run_sq_counters.sh ${tag}_cic_dec k_mix_dec_mfma tools/pmc_bank.py 3 > /dev/null
bash tools/run_sq_counters.sh ${tag}_big256_rows k_big256_rows tools/pmc_streambank.py > /dev/null
bash tools/run_sq_counters.sh ${tag}_big256_cols k_big256_cols tools/pmc_streambank.py > /dev/null
bash tools/run_sq_counters.sh ${tag}_fastfir k_fastfir_t128 tools/pmc_streambank.py > /dev/null
# kernel timelines of the two-stage calls (two output buffers against three)
bash tools/timeline_two_stage.sh > /dev/null 2>&1 || true
rm -rf $o/${tag}_trace $o/${tag}_pmc_* $o/${tag}_fused_dec $o/${tag}_cic_dec $o/${tag}_big256_rows $o/${tag}_big256_cols $o/${tag}_fastfir
ls -la $o | grep ${tag}_
