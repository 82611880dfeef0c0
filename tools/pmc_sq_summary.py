#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc SQ counter passes (counter_collection CSVs):
    pmc_sq_summary.py file.csv [file.csv ...]
Per-dispatch means of every counter and the derived fractions quoted in DESIGN.md:
  matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
  issuing / issue-stalled / parked = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
  (all in the same unit, quad-cycles summed over waves)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "").replace("isi::", "")
    return re.sub(r"\(.*$", "", name)[:70]


def main(*paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
    order = sorted(mean, key=lambda k: -mean[k].get("GRBM_GUI_ACTIVE", 0) * len(agg[k].get("GRBM_GUI_ACTIVE", [1])))
    for k in order:
        m = mean[k]
        if m.get("GRBM_GUI_ACTIVE", 0) * len(agg[k].get("GRBM_GUI_ACTIVE", [])) < 1e6:
            continue
        n = len(next(iter(agg[k].values())))
        line = [f"{k}   ({n} dispatches)"]
        d = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            d.append(f"matrix pipe busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * m['GRBM_GUI_ACTIVE'] / 8):5.1f} %")
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            f = lambda c: 100 * m.get(c, 0.0) / wc
            d.append(f"of the waves' cycles: issuing {f('SQ_ACTIVE_INST_ANY'):4.1f} % (VALU {f('SQ_ACTIVE_INST_VALU'):4.1f} %, "
                     f"LDS {f('SQ_ACTIVE_INST_LDS'):4.1f} %), issue-stalled {f('SQ_WAIT_INST_ANY'):4.1f} % "
                     f"(LDS {f('SQ_WAIT_INST_LDS'):4.1f} %), parked {f('SQ_WAIT_ANY'):4.1f} %")
        if m.get("SQ_LDS_IDX_ACTIVE"):
            d.append(f"LDS bank conflicts {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:4.1f} % of LDS cycles")
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
            d.append(f"VALU instructions per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}")
        print(line[0])
        print("   " + "   | ".join(d))
        for c in sorted(m):
            print(f"   {c:32s} {m[c]:16.1f}")
        print()


if __name__ == "__main__":
    main(*sys.argv[1:])
