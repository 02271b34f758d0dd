# kernel-trace timelines of one factorisation: the default schedule, and with the S-preloading rows kernel + K-split narrow
# updates switched on (see scripts/r3_timeline.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/tl_a gpurun_out/tl_b
rocprofv3 --kernel-trace -d gpurun_out/tl_a -o a --output-format csv -- python3 scripts/r3_timeline_target.py wing1m > gpurun_out/tl_a.log 2>&1 &&
rocprofv3 --kernel-trace -d gpurun_out/tl_b -o b --output-format csv -- python3 scripts/r3_timeline_target.py wing1m rows_preload_wg=1024 narrow_split=4 > gpurun_out/tl_b.log 2>&1 &&
python3 scripts/r3_timeline.py "$(find gpurun_out/tl_a -name '*kernel_trace.csv' | head -n 1)" 400 > gpurun_out/r3_timeline_wing1m.txt &&
python3 scripts/r3_timeline.py "$(find gpurun_out/tl_b -name '*kernel_trace.csv' | head -n 1)" 400 > gpurun_out/r3_timeline_wing1m_preload_split.txt
rm -rf gpurun_out/tl_a gpurun_out/tl_b
head -n 3 gpurun_out/r3_timeline_wing1m.txt gpurun_out/r3_timeline_wing1m_preload_split.txt
