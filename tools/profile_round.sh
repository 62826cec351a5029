#!/bin/bash
# One measurement pass on the GPU box (run from the repository root): the artefacts profiles/README.md lists.
#   bash tools/profile_round.sh <subdir of gpurun_out>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-opt-in --no-size-table --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/bench_wgrad_f32.py layers=1,2,3,4,5,6,7,8 fwd=1 > $O/pmc_sq.log 2>&1 || exit 1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
python3 $R/tools/sq_summary.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/sq_counters.txt
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq       # the raw traces are tens of MB
echo profile pass done
