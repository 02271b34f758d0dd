run() { env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['forward_ms'],2), round(d['forward_split_ms']['assemble_factorise'],2))"; }
run FEMO_LEFT_MIN=16 FEMO_LEFT_MAX=2048
run FEMO_LEFT_MIN=8 FEMO_LEFT_MAX=2048
run FEMO_LEFT_MIN=32 FEMO_LEFT_MAX=2048
run FEMO_LEFT_MIN=64 FEMO_LEFT_MAX=2048
run FEMO_LEFT_MIN=16 FEMO_LEFT_MAX=1024
run FEMO_LEFT_MIN=16 FEMO_LEFT_MAX=5000
run FEMO_LEFT_MIN=16 FEMO_LEFT_MAX=100000
run FEMO_LEFT_MIN=32 FEMO_LOOKAHEAD_CNT=32
run FEMO_LEFT_MIN=64 FEMO_LOOKAHEAD_CNT=64
run FEMO_LEFT_MIN=16 FEMO_LOOKAHEAD_CNT=8
run FEMO_LEFT_MIN=16 FEMO_LEFT_MAX=2048
