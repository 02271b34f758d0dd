# Round-6 profiles on the GPU box (through gpurun): rocprofv3 kernel-trace statistics of the bench command, two counter passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing domains beside them) on bench.py for the per-kernel HBM bytes, SQ counter
# passes for the element operator.  The program itself follows `--`.
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r6_lib.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
for d in prof_r6 pmc_r6f pmc_r6w pmc_r6lf pmc_r6lw pmc_r6s1 pmc_r6s2 pmc_r6s3; do rm -rf gpurun_out/$d; mkdir -p gpurun_out/$d; done
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-keep-numbering-leg"
run 400 gpurun_out/prof_r6/bench_under_rocprof.json rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r6 -o r6 --output-format csv -- $BENCH
run 400 gpurun_out/pmc_r6f/out.txt rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r6f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
run 400 gpurun_out/pmc_r6w/out.txt rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r6w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
run 300 gpurun_out/pmc_r6lf/out.txt rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r6lf -o f --output-format csv -- python3 scripts/r3_pmc_target.py wing1m
run 300 gpurun_out/pmc_r6lw/out.txt rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r6lw -o w --output-format csv -- python3 scripts/r3_pmc_target.py wing1m
run 300 gpurun_out/r6_trailing_meta.txt python3 scripts/r3_pmc_target.py wing1m gpurun_out/r6_trailing_meta.json
run 300 gpurun_out/pmc_r6s1/out.txt rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d gpurun_out/pmc_r6s1 -o s --output-format csv -- python3 scripts/r3_pmc_target.py wing1m
run 300 gpurun_out/pmc_r6s2/out.txt rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_r6s2 -o s --output-format csv -- python3 scripts/r3_pmc_target.py wing1m
run 300 gpurun_out/pmc_r6s3/out.txt rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SALU -d gpurun_out/pmc_r6s3 -o s --output-format csv -- python3 scripts/r3_pmc_target.py wing1m
f() { find "$1" -name '*counter_collection.csv' | head -1; }
python3 scripts/aggregate_pmc.py "$(f gpurun_out/pmc_r6f)" "$(f gpurun_out/pmc_r6w)" gpurun_out/r6_pmc_wing1m.json wing1m
python3 scripts/r4_pmc_levels.py "$(f gpurun_out/pmc_r6lf)" "$(f gpurun_out/pmc_r6lw)" gpurun_out/r6_trailing_meta.json gpurun_out/r6_pmc_trailing_levels wing1m
python3 scripts/r4_pmc_sq_apply.py "$(f gpurun_out/pmc_r6s1)" "$(f gpurun_out/pmc_r6s2)" "$(f gpurun_out/pmc_r6s3)" > gpurun_out/r6_pmc_sq_apply.txt 2>&1
cp profiles/pmc_wing1m.json gpurun_out/pmc_wing1m.json
find gpurun_out/prof_r6 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r6_wing1m_kernel_stats.csv
cat gpurun_out/r6_pmc_sq_apply.txt; tail -n 3 gpurun_out/pmc_r6s1/out.err gpurun_out/pmc_r6s2/out.err gpurun_out/pmc_r6s3/out.err
