for mb in 0 16 32 64 128; do echo "batch $mb MB"; PEBBLEGPU_BIG_BATCH_MB=$mb python3 tools/bench_streambank.py 4 128 2>&1 | tail -1 | cut -c1-400; done
