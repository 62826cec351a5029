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
python tools/bench_configs.py > $O/configs_3_5.log 2>&1; echo "configs rc=$?"
{ for B in 128 64 32 16 8; do echo "== tools/ab_lpt.py B=$B (old: natural order, skipping from 4096 workgroups; new: full tiles first, skipping wherever the order applies; all: every tap multiplied)"; python tools/ab_lpt.py B=$B rounds=4 2>&1 | grep -v amdgpu; done; } > $O/launch_order_ab.txt 2>&1
{ echo "== tools/conv_tail.py lpt=0 (natural order)"; python tools/conv_tail.py layers=2,3,4,5,6,7,8 lpt=0 2>&1 | grep -v amdgpu; echo "== tools/conv_tail.py lpt=1 (full tiles first)"; python tools/conv_tail.py layers=2,3,4,5,6,7,8 lpt=1 2>&1 | grep -v amdgpu;
  echo "== tools/conv_tail.py which=wgrad"; python tools/conv_tail.py which=wgrad layers=2,3,4,5,6,7,8 2>&1 | grep -v amdgpu; } > $O/conv_tail.txt 2>&1
{ echo "== tools/ab_wgrad.py arms=a3,a0 (a3 = no wave priority by remaining work, a0 = default)"; python tools/ab_wgrad.py arms=a3,a0 rounds=5 2>&1 | grep -v amdgpu;
  echo "== tools/wgrad_spread.py layers=3,8 (default: priorities on)"; python tools/wgrad_spread.py layers=3,8 2>&1 | grep -v amdgpu;
  echo "== tools/ab_wgrad.py arms=l96,l128,l160,l192"; python tools/ab_wgrad.py arms=l96,l128,l160,l192 rounds=4 layers=2,3,4,5,6,7,8 2>&1 | grep -v amdgpu; } > $O/wgrad_priority_ab.txt 2>&1
echo final pass done
