cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 300 gpurun_out/r4y_assemble_variants.txt python scripts/r4_assemble_variants.py
cat gpurun_out/r4y_assemble_variants.txt
bash scripts/r4_run28.sh
