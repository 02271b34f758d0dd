"""Per-launch HBM traffic of the rank-k updates of ONE factorisation against their compulsory bytes, level by level.
    python scripts/r3_pmc_levels.py FETCH_counter_collection.csv WRITE_counter_collection.csv meta.json out_prefix
meta.json: scripts/r3_pmc_target.py (instrumented run: level, time, algorithmic flops and compulsory bytes per launch, in launch
order); the counter files: the same program under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per dispatch; gfx950 correction
2 * FETCH + WRITE, MI355X_MICROARCH.md).  The last len(meta) k_trailing_mfma dispatches of each pass are the target's last factorisation."""
import csv, json, sys
from collections import defaultdict


def trailing(path, counter):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and "k_trailing_mfma" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def main():
    fetch, write, meta, prefix = sys.argv[1:5]
    launches = json.load(open(meta))["launches"]
    n = len(launches)
    fv, wv = trailing(fetch, "FETCH_SIZE")[-n:], trailing(write, "WRITE_SIZE")[-n:]
    assert len(fv) == n and len(wv) == n, (len(fv), len(wv), n)
    lev = defaultdict(lambda: dict(launches=0, us=0.0, flops=0.0, compulsory=0.0, traffic=0.0, fronts=0))
    for L, f, w in zip(launches, fv, wv):
        a = lev[L["level"]]
        a["launches"] += 1; a["us"] += L["us"]; a["flops"] += L["flops"]; a["compulsory"] += L["compulsory_bytes"]
        a["traffic"] += (2 * f + w) * 1024; a["fronts"] = L["fronts"]
    tot = dict(us=sum(a["us"] for a in lev.values()), compulsory=sum(a["compulsory"] for a in lev.values()), traffic=sum(a["traffic"] for a in lev.values()),
               flops=sum(a["flops"] for a in lev.values()))
    out = dict(workload=json.load(open(meta))["workload"], correction="(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes", levels={str(k): v for k, v in sorted(lev.items())}, total=tot)
    json.dump(out, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".md", "w") as fh:
        fh.write("| level | fronts | launches | time (us) | GFLOP | TFLOP/s | compulsory MB | counter MB | counter / compulsory | counter TB/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, a in sorted(lev.items()):
            fh.write(f"| {k} | {a['fronts']} | {a['launches']} | {a['us']:.0f} | {a['flops'] / 1e9:.1f} | {a['flops'] / a['us'] / 1e6:.1f} | {a['compulsory'] / 1e6:.0f} | "
                     f"{a['traffic'] / 1e6:.0f} | {a['traffic'] / a['compulsory']:.2f} | {a['traffic'] / a['us'] / 1e6:.2f} |\n")
        fh.write(f"| all | | {n} | {tot['us']:.0f} | {tot['flops'] / 1e9:.1f} | {tot['flops'] / tot['us'] / 1e6:.1f} | {tot['compulsory'] / 1e6:.0f} | {tot['traffic'] / 1e6:.0f} | "
                 f"{tot['traffic'] / tot['compulsory']:.2f} | {tot['traffic'] / tot['us'] / 1e6:.2f} |\n")
    print(open(prefix + ".md").read())


if __name__ == "__main__":
    main()
