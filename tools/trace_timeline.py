#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/bench_step.py: the time line of ONE steady training step (from one crop_kernel to the next):
per launch its start (us after the step's first), duration, queue and name -- which launches of the two-stream backward pass overlap.
    python tools/trace_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
crops = [i for i, e in enumerate(ev) if "crop_kernel" in e[3]]
a, b = crops[-back - 1], crops[-back]
t0 = ev[a][0]
queues = sorted({e[2] for e in ev[a:b]})
print("step of %d launches, %.1f us; queues %s" % (b - a, (ev[b][0] - t0) / 1e3, queues))
busy_until = {q: 0 for q in queues}
for s, e, q, n in ev[a:b]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = (n.split("<")[0] + ("<" + n.split("<", 1)[1].split(">")[0][:24] + ">" if "<" in n else "")).split("(")[0][:48]
    others = [qq for qq in queues if qq != q and busy_until[qq] > s]
    print("%9.1f %8.1f  q%-3s %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, queues.index(q), "||" if others else "  ", n))
    busy_until[q] = max(busy_until[q], e)
