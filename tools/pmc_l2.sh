R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && export SPLIT_CASES=31 && timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/l2 -- python3 $R/tools/bench_split.py layers=8 > $R/gpurun_out/l2.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/l2/**/*counter_collection.csv", recursive=True)
if not f: print(open("$R/gpurun_out/l2.log").read()[-1500:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-48:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    if "conv" in k or "wgrad" in k:
        us = sum(d["us"]) / len(d["us"])
        print(k, "%.0f us" % us, {c: "%.3e" % (sum(v) / len(v)) for c, v in d.items() if c != "us"})
PY
rm -rf $R/gpurun_out/l2
