#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc SQ_... GRBM_GUI_ACTIVE --kernel-trace` pass per (kernel, grid): launch time, the clock the chip
held (GRBM_GUI_ACTIVE is summed over the 8 XCDs), MFMA-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs) and the split
of wave time into parked / issue-stalled / issuing (quad-cycle counters; MI355X_MICROARCH.md, rocprofv3 PMC slots).

    python tools/sq_summary.py <counter_collection.csv> [name filter, default 'wgrad|conv_igemm|conv_dma'] > profiles/rNN/sq_*.txt
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return name.split("(")[0]


def main(path, pat="wgrad|conv_igemm|conv_dma"):
    rx = re.compile(pat)
    disp = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not rx.search(k):
            continue
        d = disp.setdefault(r["Dispatch_Id"], dict(kernel=k, grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]),
                                                   us=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, c={}))
        d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    groups = collections.OrderedDict()
    for d in disp.values():
        groups.setdefault((d["kernel"], d["grid"] // d["wg"]), []).append(d)
    for (k, wgs), ds in groups.items():
        n = len(ds)
        us = sum(d["us"] for d in ds) / n
        c = {name: sum(d["c"].get(name, 0.0) for d in ds) / n for name in ds[0]["c"]}
        ghz = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (us * 1e3) if us else 0.0
        line = "%-34s %6d workgroups x%d: %8.1f us, clock %.2f GHz" % (k, wgs, n, us, ghz)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and ghz:
            line += ", MFMA pipe busy %.1f %%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e3 * ghz * 1024))
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            line += "; of wave time: parked %.1f %%, issue-stalled %.1f %% (LDS %.1f %%), issuing %.1f %%" % (
                100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc,
                100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc)
        print(line)
        for name in sorted(c):
            print("   %-28s %.4e" % (name, c[name]))


if __name__ == "__main__":
    main(*sys.argv[1:3])
