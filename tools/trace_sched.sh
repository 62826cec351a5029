#!/bin/bash
# kernel trace + time line of one rank's step under given tools/bench_step.py arguments (r06: sched=<arm of tools/ab_wgrad_schedule.py>, two=0|1)
#   bash tools/trace_sched.sh <subdir of gpurun_out> B S tag [bench_step args ...]          e.g.  ... r06 16 64 L1d sched=L1d     ... r06 128 64 two1 two=1
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
B=${2:-16}; S=${3:-64}; C=${4:-p}; shift 4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_${B}_${S}_$C -- python3 $R/tools/bench_step.py B=$B S=$S steps=30 "$@" > $O/trace_sched_${B}_${S}_$C.log 2>&1 || exit 1
python3 $R/tools/trace_timeline.py $(find $O/tr_${B}_${S}_$C -name "*kernel_trace.csv" | head -1) > $O/step_timeline_B${B}_S${S}_$C.txt
rm -rf $O/tr_${B}_${S}_$C
grep "ms/step" $O/trace_sched_${B}_${S}_$C.log | head -1
