#!/bin/bash
# One gpurun call: tools/stress_determinism.py on the product library and on variant libraries (tools/bin/libddsp_amd_<name>.so).
#   gpurun --timeout 900 -- 'bash tools/gpu_stress.sh <tag> <iters> [variant ...]'
TAG=${1:-r04a}; ITERS=${2:-2000}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GUID" | head -2 > $OUT/box.txt; hostname >> $OUT/box.txt
echo "== product library"
timeout 600 python tools/stress_determinism.py --iters $ITERS --label head --out $OUT/stress_head.jsonl 2>&1 | grep "MISMATCH\|CASE\|SUMMARY\|Error\|error" | cut -c1-400 | tee $OUT/stress_head.txt | tail -30
for v in "$@"; do
  echo "== variant $v"
  timeout 600 python tools/stress_determinism.py --lib tools/bin/libddsp_amd_$v.so --iters $ITERS --label $v --out $OUT/stress_$v.jsonl 2>&1 | grep "MISMATCH\|CASE\|SUMMARY\|Error\|error" | cut -c1-400 | tee $OUT/stress_$v.txt | tail -30
done
echo "== done"
