for wc in 256 512 1024 2048 4096; do FEMO_WIDE_CNT=$wc python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('WIDE_CNT', $wc, round(d['forward_ms'],2), round(d['adjoint_ms'],2), d['forward_split_ms'])"; done
for t in auto right left; do FEMO_TRAILING=$t python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TRAILING', '$t', round(d['forward_ms'],2), d['forward_split_ms'])"; done
