# usage: pmc.sh <outdir> <args...>   (run from repo root on the GPU box)
R=$GRAFT_REPO_ROOT; out=$1; shift
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/$out -- python3 $R/tools/bench_wgrad_f32.py "$@" > $R/gpurun_out/$out.log 2>&1 && python3 $R/tools/sq_summary.py $(find $R/gpurun_out/$out -name "*counter_collection.csv" | head -1) > $R/gpurun_out/$out.txt
