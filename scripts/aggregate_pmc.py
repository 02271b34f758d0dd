"""Aggregate two rocprofv3 counter passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into a
per-kernel summary: launches, average duration and HBM bytes per launch.

    python scripts/aggregate_pmc.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [workload]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes
(MI355X_MICROARCH.md, "HBM"), so the corrected figure is (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
The kernels of this library read through 8-byte-per-lane loads, an access width the guide calls
uncalibrated: treat the absolute numbers as estimates, the ratios between kernels as reliable.
With `workload` given, the summary is also written to profiles/pmc_<workload>.json in the form bench.py reads.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def read(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else None
    fa, wa = read(fetch, "FETCH_SIZE"), read(write, "WRITE_SIZE")
    rows = []
    for name, (n, fkb, us) in fa.items():
        wn, wkb, _ = wa.get(name, (0, 0.0, 0.0))
        f_per = fkb / n
        w_per = wkb / wn if wn else 0.0
        rows.append({"kernel": name, "launches": n, "avg_us": us / n, "fetch_KB": f_per, "write_KB": w_per,
                     "hbm_bytes_raw": (f_per + w_per) * 1024, "hbm_bytes_corrected": (2 * f_per + w_per) * 1024,
                     "total_ms": us * 1e-3})
    rows.sort(key=lambda r: -r["total_ms"])
    json.dump(rows, open(out, "w"), indent=1)
    if workload:
        root_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root_)
        from femo_alpha_amd import _build
        # the digest of the library sources the counters were collected on: bench.py quotes these bytes only while it matches
        d = {"source": os.path.basename(out), "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch",
             "source_digest": _build.source_digest()}
        for r in rows:
            if r["kernel"].startswith("femo::k_apply4"):
                d["apply_hbm_bytes_per_launch"] = r["hbm_bytes_corrected"]
                d["apply_kernel"] = r["kernel"]
        tr = [r for r in rows if r["kernel"].startswith("femo::k_trailing")]
        if tr:          # both template instances of the rank-k update, weighted by their launches
            n = sum(r["launches"] for r in tr)
            d["trailing_hbm_bytes_per_launch"] = sum(r["hbm_bytes_corrected"] * r["launches"] for r in tr) / n
            d["trailing_kernel"] = "femo::k_trailing_mfma (all instances)"
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        json.dump(d, open(os.path.join(root, "profiles", f"pmc_{workload}.json"), "w"), indent=1)
    for r in rows[:12]:
        print(f"{r['kernel'][:60]:60s} n={r['launches']:5d} avg {r['avg_us']:9.1f} us  hbm/launch {r['hbm_bytes_corrected'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
