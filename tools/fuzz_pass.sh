#!/bin/bash
# a long run of the randomised checkers (tests/fuzz/) with fresh seeds: bash tools/fuzz_pass.sh <subdir of gpurun_out> <first seed> [scale]
# every case prints a line into its own log (a sign of life for the box); the summary lines go to fuzz_more.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
S0=${2:-61}; K=${3:-1}
run() { name=$1; shift; echo "== python tests/fuzz/$name.py $*" >> $O/fuzz_more.txt; timeout -k 10 1100 python tests/fuzz/$name.py "$@" > $O/fuzz_$name.log 2>&1; echo "rc=$? $(grep -c '^FAIL' $O/fuzz_$name.log) FAIL lines; $(tail -1 $O/fuzz_$name.log)" >> $O/fuzz_more.txt; grep '^FAIL' $O/fuzz_$name.log | head -5 >> $O/fuzz_more.txt; }
run fuzz_ops n=$((3000 * K)) seed=$S0
run fuzz_pointwise n=$((1200 * K)) seed=$((S0 + 1))
run fuzz_patches n=$((1200 * K)) seed=$((S0 + 2))
run fuzz_nets n=$((120 * K)) seed=$((S0 + 3))
run fuzz_big n=$((250 * K)) seed=$((S0 + 4))
run fuzz_big n=$((150 * K)) seed=$((S0 + 5)) sides=32,64,128
run fuzz_split n=$((150 * K)) seed=$((S0 + 6))
cat $O/fuzz_more.txt
