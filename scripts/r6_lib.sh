# helpers of the round-5 GPU-box scripts: every step runs under its own time limit; after a step that was killed at its limit
# no further GPU step is started (a hung kernel must not be followed by more launches on the same box)
run() {   # run <seconds> <log> <command...>
    local lim=$1 log=$2 rc=0; shift 2
    timeout -k 10 "$lim" "$@" > "$log" 2> "${log%.*}.err" || rc=$?      # (|| keeps a failing step from ending a `set -e` caller here)
    echo "rc=$rc" >> "${log%.*}.err"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed at its limit ($lim s): $*; stopping"; exit 1; fi
    return 0      # an ordinary failure (rc != 0) is in the .err file; the pass goes on
}
