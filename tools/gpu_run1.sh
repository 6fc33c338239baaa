set -e
tools/run_sq_counters.sh r02_spectrum_base k_spectrum tools/pmc_spectrum.py
python3 tools/bench_configs.py 2 > gpurun_out/r02_base_c2.json
python3 tools/bench_configs.py 3 > gpurun_out/r02_base_c3.json
python3 tools/bench_streambank.py 4 128 > gpurun_out/r02_base_c4.json
cat gpurun_out/r02_base_c*.json
