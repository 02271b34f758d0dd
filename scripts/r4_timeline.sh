# kernel-trace timeline of one factorisation on the final sources of round 4 (scripts/r3_timeline.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/tl_r4
rocprofv3 --kernel-trace -d gpurun_out/tl_r4 -o a --output-format csv -- python3 scripts/r3_timeline_target.py wing1m > gpurun_out/tl_r4.log 2>&1 &&
python3 scripts/r3_timeline.py "$(find gpurun_out/tl_r4 -name '*kernel_trace.csv' | head -n 1)" 600 > gpurun_out/r4_timeline_wing1m.txt
rm -rf gpurun_out/tl_r4
head -n 24 gpurun_out/r4_timeline_wing1m.txt
