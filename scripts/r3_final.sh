# Round-3 measurement pass on the GPU box: tests, bench lines (1 GPU; 2 / 4 ranks sharing the card as a rehearsal of the launch
# path), per-level profiles, the strong-scaling composition, then the rocprof passes (scripts/r3_rocprof.sh).
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r3_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_final_tests.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3_smoke.log
./scripts/micro/wsum_micro > gpurun_out/r3_micro.txt 2>&1; ./scripts/micro/stage_qpoints_micro >> gpurun_out/r3_micro.txt 2>&1
# the counter passes first: they write profiles/pmc_wing1m.json with the digest of these sources, which the bench lines below read
bash scripts/r3_rocprof.sh > gpurun_out/r3_rocprof.log 2>&1
python bench.py --steps 40 --warmup 5 > gpurun_out/r3_bench_wing1m.json 2> gpurun_out/r3_bench_wing1m.err
FEMO_BENCH_WORKLOAD=plate250k python bench.py --steps 40 --warmup 5 > gpurun_out/r3_bench_plate250k.json 2> gpurun_out/r3_bench_plate250k.err
python bench.py --steps 20 --warmup 3 --nquad 5 --no-cpu-baseline --no-keep-numbering-leg > gpurun_out/r3_bench_wing1m_nquad5.json 2> gpurun_out/r3_bench_wing1m_nquad5.err
python bench.py --gpus 2 --share-gpu --steps 10 --warmup 2 > gpurun_out/r3_bench_dist2.json 2> gpurun_out/r3_bench_dist2.err
python bench.py --gpus 4 --share-gpu --steps 10 --warmup 2 > gpurun_out/r3_bench_dist4.json 2> gpurun_out/r3_bench_dist4.err
python scripts/r2_levels.py wing1m > gpurun_out/r3_levels_wing1m.txt 2>&1
python scripts/r3_sweeps.py wing1m > gpurun_out/r3_sweeps_wing1m.txt 2>&1
bash scripts/r3_timeline.sh > gpurun_out/r3_timeline.log 2>&1
python scripts/r3_amdahl.py wing1m > gpurun_out/r3_amdahl_wing1m.json 2> gpurun_out/r3_amdahl_wing1m.md
python scripts/bench_csr.py c3 > gpurun_out/r3_csr_wing1m.txt 2>&1
python scripts/bench_dynamic.py 82 410 100 gpurun_out/r3_dynamic_500k.json > gpurun_out/r3_dynamic_500k.log 2>&1
FEMO_BENCH_WORKLOAD=wing4m python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-keep-numbering-leg > gpurun_out/r3_bench_wing4m.json 2> gpurun_out/r3_bench_wing4m.err
echo final rc=$?
