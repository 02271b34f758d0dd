cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4z_pack.txt python scripts/r4_pack_ab.py
cat gpurun_out/r4z_pack.txt; tail -3 gpurun_out/r4z_pack.err
