#!/bin/bash
# kernel trace of one rank's step at the per-rank batch: where the time of a small step goes (kernel time against the gaps between kernels)
#   bash tools/trace_small.sh <subdir of gpurun_out> B S [comm]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B=${2:-16}; S=${3:-25}; C=${4:-none}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_${B}_${S}_$C -- python3 $R/tools/bench_step.py B=$B S=$S steps=50 comm=$C > $O/trace_${B}_${S}_$C.log 2>&1 || exit 1
cp $(find $O/tr_${B}_${S}_$C -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${B}_${S}_$C.csv
python3 $R/tools/trace_gaps.py $(find $O/tr_${B}_${S}_$C -name "*kernel_trace.csv" | head -1) > $O/gaps_${B}_${S}_$C.txt
rm -rf $O/tr_${B}_${S}_$C
tail -3 $O/trace_${B}_${S}_$C.log
