"""Per-dispatch SQ counters of the rank-k updates of the last factorisation in a rocprofv3 --pmc run of scripts/r3_pmc_target.py:
occupancy (wave cycles per busy cycle), share of wave cycles spent waiting, MFMA-busy share.
    python3 scripts/r3_pmc_sq.py <counter_collection.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "k_trailing_mfma" not in r["Kernel_Name"]:
        continue
    d = by.setdefault(int(r["Dispatch_Id"]), {"name": "gather" if "<true>" in r["Kernel_Name"] or "ILb1" in r["Kernel_Name"] else "plain",
                                              "grid": r.get("Grid_Size", r.get("Grid_Size_X", ""))})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(by)[-67:]                      # the last factorisation's 67 launches
print("launch kind   grid        waves   wave-cycles/busy-cycle  wait/wave-cycles  mfma-busy/busy  gui-active")
for k, i in enumerate(ids):
    d = by[i]
    wc, bc = d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_BUSY_CYCLES", 1)
    print(f"{k:3d} {d['name']:6s} {str(d['grid']):>10s} {d.get('SQ_WAVES', 0):9.0f} {wc / max(bc, 1):10.2f} {d.get('SQ_WAIT_INST_ANY', 0) / max(wc, 1):14.2f} "
          f"{d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(bc, 1):12.2f} {d.get('GRBM_GUI_ACTIVE', 0):12.0f}")
