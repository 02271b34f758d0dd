cd $GRAFT_REPO_ROOT
for o in "$@"; do
  n=$(echo $o | tr ' =' '__')
  timeout -k 10 200 python scripts/r3_sweeps.py wing1m $o > gpurun_out/sw_$n.txt 2>&1 || exit 1
  echo "$o: $(tail -n 1 gpurun_out/sw_$n.txt)"
done
