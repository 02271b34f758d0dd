"""SQ counters of the matrix-free element operator k_apply4 from a rocprofv3 --pmc run (scripts/r4_rocprof.sh): what the kernel
waits for.  Per dispatch, averaged over the launches: waves, wave-cycles per busy cycle (resident waves per SIMD-ish), the share of
wave cycles spent waiting for any instruction dependency, VALU instructions issued, the share of busy cycles in which the vector
ALU executes, LDS instructions and LDS bank-conflict cycles.
    python3 scripts/r4_pmc_sq_apply.py <counter_collection.csv> [<more.csv> ...]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "k_apply4" not in name and "k_gather_sum" not in name:
            continue
        key = "k_apply4" if "k_apply4" in name else "k_gather_sum"
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
for key in acc:
    print(key)
    a = {k: acc[key][k] / max(cnt[key][k], 1) for k in acc[key]}
    for k in sorted(a):
        print(f"    {k:32s} {a[k]:16.1f}   (average of {cnt[key][k]} dispatches)")
    wc, bc = a.get("SQ_WAVE_CYCLES", 0.0), a.get("SQ_BUSY_CYCLES", 0.0)
    if wc and bc:
        print(f"    wave cycles / busy cycle         {wc / bc:16.2f}")
    if wc and "SQ_WAIT_INST_ANY" in a:
        print(f"    waiting share of wave cycles     {a['SQ_WAIT_INST_ANY'] / wc:16.2f}")
    if wc and "SQ_ACTIVE_INST_VALU" in a:
        print(f"    VALU-executing share of wave cyc {a['SQ_ACTIVE_INST_VALU'] / wc:16.2f}")
    if "SQ_INSTS_VALU" in a and "SQ_WAVES" in a:
        print(f"    VALU instructions per wave       {a['SQ_INSTS_VALU'] / max(a['SQ_WAVES'], 1):16.1f}")
