#!/bin/bash
# kernel trace + time line of one rank's step under a filter-gradient schedule of tools/ab_wgrad_schedule.py (r06 experiment)
#   bash tools/trace_sched.sh <subdir of gpurun_out> B S sched
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B=${2:-16}; S=${3:-64}; C=${4:-p}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_${B}_${S}_$C -- python3 $R/tools/bench_step.py B=$B S=$S steps=50 sched=$C > $O/trace_sched_${B}_${S}_$C.log 2>&1 || exit 1
python3 $R/tools/trace_timeline.py $(find $O/tr_${B}_${S}_$C -name "*kernel_trace.csv" | head -1) > $O/step_timeline_B${B}_S${S}_$C.txt
rm -rf $O/tr_${B}_${S}_$C
grep "ms/step" $O/trace_sched_${B}_${S}_$C.log | head -1
