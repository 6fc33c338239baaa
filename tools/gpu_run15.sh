timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "config5" 2>&1 | tail -3
python3 tools/bench_streambank.py 4 128 2>&1 | tail -1 | cut -c1-420
PEBBLEGPU_BIG_ROWS_WAVE=1 python3 tools/bench_streambank.py 4 128 2>&1 | tail -1 | cut -c1-420
