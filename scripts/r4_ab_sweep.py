"""One-at-a-time sweep of the schedule thresholds on a workload other than the one they were tuned on (wing1m): every set spells
out all swept keys, one of them away from its default; scripts/r4_ab.py does the interleaved timing."""
import os, subprocess, sys
base = dict(rows_fine_wg=96, narrow_fine_wg=128, diag_v1_cnt=512, super_panel=512, super_panel_cnt=64, left_min=64, left_max=2048,
            lookahead=1, lookahead_cnt=16, fuse_rows_cnt=4096, xinv_small_cnt=32)
alts = dict(rows_fine_wg=[0, 32, 256], narrow_fine_wg=[0, 32, 512], diag_v1_cnt=[128, 2048], super_panel=[256, 0], super_panel_cnt=[16, 256],
            left_min=[16, 256], left_max=[512, 8192], lookahead=[0], lookahead_cnt=[4, 64], fuse_rows_cnt=[1024, 100000], xinv_small_cnt=[8, 128])
sets = [dict(base)]
for k, vs in alts.items():
    for v in vs:
        s = dict(base); s[k] = v
        sets.append(s)
args = [",".join(f"{k}={v}" for k, v in s.items()) for s in sets]
here = os.path.dirname(os.path.abspath(__file__))
r = subprocess.run([sys.executable, os.path.join(here, "r4_ab.py"), sys.argv[1], *args], capture_output=True, text=True)
lines = [l for l in r.stdout.splitlines() if "factor_ms" in l]
ref = float(lines[0].split("median")[1].split()[0])
print(f"{sys.argv[1]}: defaults {lines[0].split(': factor_ms')[1]}")
for s, l in zip(sets[1:], lines[1:]):
    k = [k for k in s if s[k] != base[k]][0]
    med = float(l.split("median")[1].split()[0])
    print(f"  {k:16s} = {s[k]:<7} median {med:.3f} ms  ({(med / ref - 1) * 100:+.1f} %)")
if r.returncode:
    print(r.stderr[-2000:])
