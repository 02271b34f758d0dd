# kernel-trace timeline of one factorisation on the final sources of round 4 (scripts/r3_timeline.py)
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r4_lib.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/tl_r4
run 300 gpurun_out/tl_r4.log rocprofv3 --kernel-trace -d gpurun_out/tl_r4 -o a --output-format csv -- python3 scripts/r3_timeline_target.py wing1m
python3 scripts/r3_timeline.py "$(find gpurun_out/tl_r4 -name '*kernel_trace.csv' | head -n 1)" 600 > gpurun_out/r4_timeline_wing1m.txt
rm -rf gpurun_out/tl_r4
head -n 24 gpurun_out/r4_timeline_wing1m.txt
