cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for w in wing1m plate250k wing4m; do
run 600 gpurun_out/r4v_ab_$w.txt python scripts/r4_ab.py $w "fuse_rows_np=128,fuse_rows_cnt=4096" "fuse_rows_np=256,fuse_rows_cnt=4096" "fuse_rows_np=256,fuse_rows_cnt=2048" "fuse_rows_np=256,fuse_rows_cnt=1024" "fuse_rows_np=384,fuse_rows_cnt=1024"
cat gpurun_out/r4v_ab_$w.txt
done
run 900 gpurun_out/r4v_fuzz.txt python scripts/fuzz_schedules.py 77 200
tail -1 gpurun_out/r4v_fuzz.txt
run 600 gpurun_out/r4v_tests.log python -m pytest tests/test_gpu_schedules.py -q -m gpu -x
tail -3 gpurun_out/r4v_tests.log
