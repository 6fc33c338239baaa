mkdir -p gpurun_out
cp build_exp/lib_TIMING.so pebblesdr_amd/libpebblegpu.so
PEBBLEGPU_FUSED_L=256 timeout -k 10 120 python3 tools/pmc_bank.py 2 > gpurun_out/tp_timing_256.txt 2>&1
PEBBLEGPU_FUSED_L=128 timeout -k 10 120 python3 tools/pmc_bank.py 2 > gpurun_out/tp_timing_128.txt 2>&1
wc -l gpurun_out/tp_timing_*.txt
