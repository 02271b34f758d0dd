cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for leaf in 12 16 24 32; do
run 600 gpurun_out/r4s_tri_$leaf.txt python scripts/r4_tri.py $leaf
cat gpurun_out/r4s_tri_$leaf.txt
done
