set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2 gpurun_out/pmc_r2f gpurun_out/pmc_r2w
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2 -o r2 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-keep-numbering-leg > gpurun_out/prof_r2/bench_under_rocprof.json 2> gpurun_out/prof_r2/rocprof.log
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_r2f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg > /dev/null 2> gpurun_out/pmc_r2f/log.txt
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_r2w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg > /dev/null 2> gpurun_out/pmc_r2w/log.txt
ls -la gpurun_out/prof_r2 gpurun_out/pmc_r2f gpurun_out/pmc_r2w
python3 scripts/aggregate_pmc.py $(ls gpurun_out/pmc_r2f/*counter_collection.csv | head -1) $(ls gpurun_out/pmc_r2w/*counter_collection.csv | head -1) gpurun_out/r2_pmc_wing1m.json wing1m
