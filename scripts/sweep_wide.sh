for wc in 0 64 256 512 1024; do FEMO_WIDE_CNT=$wc python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('WIDE_CNT', $wc, d['value'], d['forward_ms'], d['adjoint_ms'], d['forward_split_ms'])"; done
