# Round-3 profiles on the GPU box (run through gpurun): rocprofv3 kernel-trace statistics of the bench command, two counter
# passes (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing domains beside them) on bench.py for the per-kernel averages, and the
# same two passes on a one-factorisation target for the per-level table of the rank-k updates.  The program itself follows `--`.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r3 gpurun_out/pmc_r3f gpurun_out/pmc_r3w gpurun_out/pmc_r3lf gpurun_out/pmc_r3lw
mkdir -p gpurun_out/prof_r3 gpurun_out/pmc_r3f gpurun_out/pmc_r3w gpurun_out/pmc_r3lf gpurun_out/pmc_r3lw
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3 -o r3 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-keep-numbering-leg > gpurun_out/prof_r3/bench_under_rocprof.json 2> gpurun_out/prof_r3/rocprof.log
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r3f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg > /dev/null 2> gpurun_out/pmc_r3f/log.txt
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r3w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg > /dev/null 2> gpurun_out/pmc_r3w/log.txt
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r3lf -o f --output-format csv -- python3 scripts/r3_pmc_target.py wing1m > /dev/null 2> gpurun_out/pmc_r3lf/log.txt
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r3lw -o w --output-format csv -- python3 scripts/r3_pmc_target.py wing1m > /dev/null 2> gpurun_out/pmc_r3lw/log.txt
python3 scripts/r3_pmc_target.py wing1m gpurun_out/r3_trailing_meta.json
f() { find "$1" -name '*counter_collection.csv' | head -1; }
python3 scripts/aggregate_pmc.py "$(f gpurun_out/pmc_r3f)" "$(f gpurun_out/pmc_r3w)" gpurun_out/r3_pmc_wing1m.json wing1m
python3 scripts/r3_pmc_levels.py "$(f gpurun_out/pmc_r3lf)" "$(f gpurun_out/pmc_r3lw)" gpurun_out/r3_trailing_meta.json gpurun_out/r3_pmc_trailing_levels
cp profiles/pmc_wing1m.json gpurun_out/pmc_wing1m.json
find gpurun_out/prof_r3 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r3_wing1m_kernel_stats.csv
ls -la gpurun_out/prof_r3 gpurun_out/pmc_r3f gpurun_out/pmc_r3lf
