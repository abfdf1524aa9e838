#!/bin/bash
# harm_table_kernel: how the eight S-wavefronts share phase B (interpolation) and phase A (row preparation)
OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT
for NB in 8 4 5 6; do
  if [ $NB = 8 ]; then unset DDSP_EXP_TABLE_NB; else export DDSP_EXP_TABLE_NB=$NB; fi
  echo "== NB=$NB"
  timeout 120 python tools/exp_table.py 32 128 2>&1 | tail -2 | cut -c1-150 | tee $OUT/harm_table_nb_$NB.json
  timeout 120 python tools/exp_table_timeline.py 32 2>&1 | grep -A40 "launch 2" | grep "tick   [34]" | cut -c1-170 | tee $OUT/timeline_harm_table_nb_$NB.txt
done
