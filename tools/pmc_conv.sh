# FETCH_SIZE of the forward kernel under a K-loop order: pmc_conv.sh <outdir> <korder>   (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT; out=$1
cd /tmp && export TMPDIR=/tmp && DRS_CONV_KORDER=$2 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$out -- python3 $R/tools/bench_wgrad_f32.py layers=6,8 fwd=1 reps=2 > $R/gpurun_out/$out.log 2>&1 && python3 - <<PY > $R/gpurun_out/$out.txt
import csv, glob, collections
f = glob.glob("$R/gpurun_out/$out/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "conv_dma" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "launches", len(v), "FETCH_SIZE x2 (gfx950) = %.3f GB per launch" % (2 * sum(v) / len(v) * 1024 / 1e9))
PY
