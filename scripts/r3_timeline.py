"""Dispatch timeline of the LAST factorisation in a rocprofv3 kernel trace (scripts/r3_timeline_target.py): for every launch
its start relative to the factorisation's first launch, its duration and the idle gap before it on the device -- what the
chain of latency-bound launches at the top of the elimination tree really costs, without the instrumented mode's event pairs.
    python3 scripts/r3_timeline.py <kernel_trace.csv> [last N launches]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
zs = [i for i, n in enumerate(names) if "k_zero_fronts" in n]
rows = rows[zs[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    mm = re.search(r"femo::(k_\w+)(<[^>]*>)?", n)
    return (mm[1] + (mm[2] or "")) if mm else n[:30]
end = t0
tot = {}
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0))
    out.append((short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3, g))
    a = tot.setdefault(out[-1][0], [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    end = max(end, e)
print(f"last factorisation: {len(rows)} launches, {(end - t0) / 1e3:.1f} us from first start to last end")
busy = sum(o[2] for o in out)
print(f"sum of kernel durations {busy:.1f} us (overlap counts twice)")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {n:5d} launches {us:10.1f} us")
print(f"--- last {nlast} launches: name, start us, duration us, gap before (negative: overlaps the previous), grid")
for o in out[-nlast:]:
    print(f"{o[0]:40s} {o[1]:10.1f} {o[2]:8.1f} {o[3]:8.1f}   {o[4]}")
