#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: per kernel name the launches, the summed run time and the summed idle gap BEFORE each launch
(start - previous end on the device time line, all streams merged; overlapping kernels give a gap of 0)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the steady part: the last 60 % of the trace
ev = ev[int(len(ev) * 0.4):]
run, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
last_end = ev[0][0]
for s, e, n in ev:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = (n.split("<")[0] + ("<" + n.split("<", 1)[1].split(">")[0][:28] + ">" if "<" in n else "")).split("(")[0][:70]
    run[n] += e - s
    gap[n] += max(0, s - last_end)
    cnt[n] += 1
    last_end = max(last_end, e)
span = ev[-1][1] - ev[0][0]
tr, tg = sum(run.values()), sum(gap.values())
print("span %.3f ms: kernels %.3f ms (%.1f %%), idle gaps %.3f ms (%.1f %%), %d launches" % (span / 1e6, tr / 1e6, 100.0 * tr / span, tg / 1e6, 100.0 * tg / span, len(ev)))
print("%-70s %7s %10s %10s %9s %9s" % ("kernel", "calls", "run us", "gap us", "avg run", "avg gap"))
for n, _ in sorted(run.items(), key=lambda kv: -(kv[1] + gap[kv[0]])):
    print("%-70s %7d %10.1f %10.1f %9.2f %9.2f" % (n, cnt[n], run[n] / 1e3, gap[n] / 1e3, run[n] / 1e3 / cnt[n], gap[n] / 1e3 / cnt[n]))
