mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipebblesdr_amd/csrc tools/ubench/fft_core.hip -o gpurun_out/fft_core 2>/dev/null && ./gpurun_out/fft_core > gpurun_out/r02_fft_core.txt; cat gpurun_out/r02_fft_core.txt
python3 tools/diag/agc_am.py > gpurun_out/r02_agc_am.txt 2>&1; cat gpurun_out/r02_agc_am.txt
