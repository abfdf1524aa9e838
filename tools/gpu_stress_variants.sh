#!/bin/bash
# gpurun -- 'bash tools/gpu_stress_variants.sh <tag> <iters> <cases> <variant> ...'   (variant "head" = the product library)
TAG=$1; ITERS=$2; CASES=$3; shift; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  LIB=""; [ "$v" != head ] && LIB="--lib tools/bin/libddsp_amd_$v.so"
  echo "== $v"
  timeout 300 python tools/stress_determinism.py $LIB --iters $ITERS --label $v --cases $CASES --out $OUT/stress_$v.jsonl 2>&1 | grep "CASE\|SUMMARY\|rror" | cut -c1-200
done
