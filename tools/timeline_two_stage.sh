# kernel timelines of configs[2] two-stage calls: round 3's first form (two output buffers, nap, queued wait) and the final one
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
o=gpurun_out/r03_timeline_two_stage.txt
echo "# tools/timeline_two_stage.sh: rocprofv3 --kernel-trace of python3 tools/ab_bank_pipe.py 2 100, the last launches (us; gap = start - end of the latest earlier kernel)" > $o
echo "## two output buffers, 8 us nap, wait packet (PEBBLEGPU_BANK_PIPE_BUFS=2 PEBBLEGPU_BANK_PIPE_HOSTWAIT=0)" >> $o
rm -rf gpurun_out/tlA; PEBBLEGPU_BANK_PIPE_BUFS=2 PEBBLEGPU_BANK_PIPE_HOSTWAIT=0 rocprofv3 --kernel-trace -d gpurun_out/tlA -- python3 tools/ab_bank_pipe.py 2 100 > gpurun_out/tlA.log 2>&1
grep "ms per call" gpurun_out/tlA.log >> $o; python3 tools/trace_timeline.py gpurun_out/tlA 22 >> $o; rm -rf gpurun_out/tlA
echo "## three output buffers, no nap, the host waits (defaults)" >> $o
rm -rf gpurun_out/tlB; rocprofv3 --kernel-trace -d gpurun_out/tlB -- python3 tools/ab_bank_pipe.py 2 100 > gpurun_out/tlB.log 2>&1
grep "ms per call" gpurun_out/tlB.log >> $o; python3 tools/trace_timeline.py gpurun_out/tlB 22 >> $o; rm -rf gpurun_out/tlB
echo "## the same without the profiler" >> $o
PEBBLEGPU_BANK_PIPE_BUFS=2 PEBBLEGPU_BANK_PIPE_HOSTWAIT=0 python3 tools/ab_bank_pipe.py 2 400 | tail -1 | sed 's/^/two buffers: /' >> $o
python3 tools/ab_bank_pipe.py 2 400 | tail -1 | sed 's/^/defaults:    /' >> $o
PEBBLEGPU_BANK_PIPE_BUFS=2 PEBBLEGPU_BANK_PIPE_HOSTWAIT=0 python3 tools/ab_bank_pipe.py 3 400 | tail -1 | sed 's/^/two buffers: /' >> $o
python3 tools/ab_bank_pipe.py 3 400 | tail -1 | sed 's/^/defaults:    /' >> $o
