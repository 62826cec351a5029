#!/bin/bash
# the measurement pass a round's profiles/ come from (one box, the final revision): bash tools/final_pass.sh <subdir of gpurun_out>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/final_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/final_gpu_tests.log; tail -2 $O/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
DRS_FORCE_COLLECTIVES=1 python bench.py --steps 40 --no-cpu-baseline --no-opt-in --no-size-table > $O/bench_forced_rccl_world1.json 2> $O/bench_forced.err; echo "forced rc=$?"
bash tools/profile_round.sh $1 > $O/profile_round.log 2>&1; echo "profile rc=$?"
cd $R
# (round 6: BASELINE configs 2-5 are in the bench line itself -- extra.configs; the launch-order / tail / priority A/Bs of rounds 4-5 live in
#  profiles/r04, profiles/r05 and are not repeated)
{ echo "# tools/soak.py at the round's final revision: two identical runs compared bit for bit";
  python tools/soak.py B=16 steps=3000 2>&1 | grep -v amdgpu;
  python tools/soak.py B=32 steps=800 lo=33 hi=100 net=dilated_icpr_rate6_densely channels=4 classes=2 2>&1 | grep -v amdgpu;
  python tools/soak.py B=16 steps=600 comm=rccl 2>&1 | grep -v amdgpu;
  echo "# tools/poison_check.py: the same step with every scratch buffer NaN / 0xFF-filled first";
  python tools/poison_check.py 2>&1 | grep -v amdgpu; python tools/poison_check.py B=16 S=64 2>&1 | grep -v amdgpu; } > $O/soak_determinism.txt 2>&1; tail -2 $O/soak_determinism.txt
echo final pass done
