export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/r02_kt; rocprofv3 --kernel-trace --stats -d gpurun_out/r02_kt -- python3 bench.py --steps 10 --no-cpu-baseline > gpurun_out/r02_kt_bench.json 2> gpurun_out/r02_kt.err; echo "rc=$?"
python3 tools/kernel_stats.py gpurun_out/r02_kt gpurun_out/r02_p_kernel_stats.csv | cut -c1-160
