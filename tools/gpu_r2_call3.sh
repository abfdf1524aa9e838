#!/bin/bash
# harm_table_kernel: four S-wavefronts with four tiles each (experiment) against eight with two
OUT=gpurun_out/${1:-r02i}; mkdir -p $OUT
for S4 in 0 1; do
  if [ $S4 = 1 ]; then export DDSP_EXP_TABLE_S4=1; else unset DDSP_EXP_TABLE_S4; fi
  echo "== S4=$S4"
  timeout 120 python tools/exp_table.py 32 128 2>&1 | tail -2 | cut -c1-260 | tee $OUT/harm_table_s4_$S4.json
  timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | grep "tick   [3-5]" | tee $OUT/timeline_harm_table_s4_$S4.txt
done
