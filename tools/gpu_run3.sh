set -e
export PEBBLEGPU_SPECTRUM_SHARED=1
for st in 0 1 2 3 4 6 8 12; do
  echo "stagger $st"; PEBBLEGPU_T128_STAGGER=$st python3 tools/bench_spectrum_sizes.py 8192 2>&1 | tail -1
done
echo "pad lds (1 WG/CU)"; PEBBLEGPU_T128_PADLDS=16384 python3 tools/bench_spectrum_sizes.py 8192 2>&1 | tail -1
