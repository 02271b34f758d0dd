# A/B of schedule options on one box: the schedule tests, then scripts/r2_levels.py once per option set (arguments, quoted)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_schedules.py -x -q > gpurun_out/ab_tests.log 2>&1; tail -n 3 gpurun_out/ab_tests.log
for o in "$@"; do
  n=$(echo $o | tr ' =' '__')
  timeout -k 10 200 python scripts/r2_levels.py wing1m $o > gpurun_out/ab_$n.txt 2>&1 || exit 1
  echo "$o: $(tail -n 1 gpurun_out/ab_$n.txt)"
done
