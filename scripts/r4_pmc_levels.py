"""Per-launch HBM traffic of the rank-k updates (k_trailing_mfma and k_trailing_fine) of ONE factorisation against their compulsory
bytes: level by level, and by the roof that binds a launch (flops / compulsory bytes above or below the chip's ridge of 9.8).
    python scripts/r4_pmc_levels.py FETCH_counter_collection.csv WRITE_counter_collection.csv meta.json out_prefix [workload]
meta.json: scripts/r3_pmc_target.py (instrumented run: level, time, algorithmic flops and compulsory bytes per launch, in launch
order); the counter files: the same program under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per dispatch; gfx950 correction
2 * FETCH + WRITE, MI355X_MICROARCH.md).  The last len(meta) rank-k dispatches of each pass are the target's last factorisation.
With ``workload``: the per-class bytes per launch are added to profiles/pmc_<workload>.json (bench.py quotes them as roofline.traffic)."""
import csv, json, os, sys
from collections import defaultdict

RIDGE = 78.6e12 / 8.0e12


def trailing(path, counter):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and "k_trailing" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def main():
    fetch, write, meta, prefix = sys.argv[1:5]
    workload = sys.argv[5] if len(sys.argv) > 5 else None
    launches = json.load(open(meta))["launches"]
    n = len(launches)
    fv, wv = trailing(fetch, "FETCH_SIZE")[-n:], trailing(write, "WRITE_SIZE")[-n:]
    assert len(fv) == n and len(wv) == n, (len(fv), len(wv), n)
    new = lambda: dict(launches=0, us=0.0, flops=0.0, compulsory=0.0, traffic=0.0, fronts=0)
    lev, cls = defaultdict(new), defaultdict(new)
    for L, f, w in zip(launches, fv, wv):
        which = "mfma_bound" if L["compulsory_bytes"] > 0 and L["flops"] / L["compulsory_bytes"] >= RIDGE else "hbm_bound"
        for a in (lev[L["level"]], cls[which]):
            a["launches"] += 1; a["us"] += L["us"]; a["flops"] += L["flops"]; a["compulsory"] += L["compulsory_bytes"]
            a["traffic"] += (2 * f + w) * 1024; a["fronts"] = L["fronts"]
    tot = {k: sum(a[k] for a in lev.values()) for k in ("us", "compulsory", "traffic", "flops")}
    out = dict(workload=json.load(open(meta))["workload"], correction="(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes",
               levels={str(k): v for k, v in sorted(lev.items())}, classes=dict(cls), total=tot)
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    row = lambda k, a: (f"| {k} | {a['fronts']} | {a['launches']} | {a['us']:.0f} | {a['flops'] / 1e9:.1f} | {a['flops'] / a['us'] / 1e6:.1f} | {a['compulsory'] / 1e6:.0f} | "
                        f"{a['traffic'] / 1e6:.0f} | {a['traffic'] / a['compulsory']:.2f} | {a['traffic'] / a['us'] / 1e6:.2f} |\n")
    with open(prefix + ".md", "w") as fh:
        fh.write("| level | fronts | launches | time (us) | GFLOP | TFLOP/s | compulsory MB | counter MB | counter / compulsory | counter TB/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, a in sorted(lev.items()):
            fh.write(row(k, a))
        for k, a in sorted(cls.items()):
            fh.write(row(k.replace("_", "-"), dict(a, fronts="")))
        fh.write(f"| all | | {n} | {tot['us']:.0f} | {tot['flops'] / 1e9:.1f} | {tot['flops'] / tot['us'] / 1e6:.1f} | {tot['compulsory'] / 1e6:.0f} | {tot['traffic'] / 1e6:.0f} | "
                 f"{tot['traffic'] / tot['compulsory']:.2f} | {tot['traffic'] / tot['us'] / 1e6:.2f} |\n")
    print(open(prefix + ".md").read())
    if workload:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "profiles", f"pmc_{workload}.json")
        d = json.load(open(path))
        for k, a in cls.items():
            d[f"trailing_{k}_hbm_bytes_per_launch"] = a["traffic"] / a["launches"]
            d[f"trailing_{k}_launches"] = a["launches"]
        json.dump(d, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
