#!/bin/bash
# rv_fft_kernel: items dealt so that an XCD takes a contiguous eighth of every round (the default) against block order
# (DDSP_EXP_RV_PLAIN_ORDER=1), in turn inside one call.
# Usage: gpurun --timeout 600 -- 'bash tools/exp_xcd_order.sh [tag]'
TAG=${1:-r05z2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for ARGS in "128 64000 48000 1" "32" "128"; do
    echo "== bench_reverb $ARGS, XCD-aware (run $rep)"
    timeout 200 python tools/bench_reverb.py $ARGS 2>&1 | tail -1 | tee -a $OUT/xcd.jsonl | cut -c1-420
    echo "== bench_reverb $ARGS, plain (run $rep)"
    DDSP_EXP_RV_PLAIN_ORDER=1 timeout 200 python tools/bench_reverb.py $ARGS 2>&1 | tail -1 | tee -a $OUT/plain.jsonl | cut -c1-420
  done
done
echo "== done"
