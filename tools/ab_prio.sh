for B in 128 16; do
for cfg in "DRS_TWO_STREAMS=0" "DRS_TWO_STREAMS=1" "DRS_TWO_STREAMS=1 DRS_WG_PRIORITY=low" "DRS_TWO_STREAMS=1 DRS_WG_PRIORITY=high"; do
  echo "== B=$B $cfg"; env $cfg python tools/bench_step.py B=$B S=64 steps=30 2>&1 | grep "ms/step "
done; done
