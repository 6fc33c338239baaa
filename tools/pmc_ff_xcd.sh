cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; o=gpurun_out; export PMC_CALIBRATE=1
for x in 0 1; do
for w in configs2 configs4; do
  case $w in configs2) script="tools/pmc_bank.py 2";; configs4) script="tools/pmc_streambank.py";; esac
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $o/x${x}_pmc_${w}_$c; PEBBLEGPU_FF_XCD=$x rocprofv3 --pmc $c --kernel-trace -d $o/x${x}_pmc_${w}_$c -- python3 $script > $o/x${x}_pmc_${w}_$c.log 2>&1
  done
  python3 tools/parse_traffic.py $o/x${x}_pmc_${w}_FETCH_SIZE $o/x${x}_pmc_${w}_WRITE_SIZE $o/x${x}_traffic_${w}.json > /dev/null
  rm -rf $o/x${x}_pmc_${w}_FETCH_SIZE $o/x${x}_pmc_${w}_WRITE_SIZE
done
done
