#!/bin/bash
# SpectralLoss: the XCD-aware block order (sl_where, csrc/spectral_loss.hip) against the plain one (DDSP_EXP_SL_PLAIN_ORDER=1),
# in turn inside one call: times, the loss's bits, and the L2s' fabric reads (FETCH_SIZE) per launch.
# Usage: gpurun --timeout 600 -- 'bash tools/exp_loss_block_order.sh [tag]'
TAG=${1:-r05z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for B in 128 32; do
    echo "== batch $B, XCD-aware order (run $rep)"
    timeout 200 python tools/bench_spectral_loss.py $B 2>&1 | tail -1 | tee -a $OUT/xcd_b$B.jsonl | cut -c1-420
    echo "== batch $B, plain order (run $rep)"
    DDSP_EXP_SL_PLAIN_ORDER=1 timeout 200 python tools/bench_spectral_loss.py $B 2>&1 | tail -1 | tee -a $OUT/plain_b$B.jsonl | cut -c1-420
  done
done
echo "== the loss's bits, both orders (batch 9: units not a multiple of 8)"
cat > /tmp/bits.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import ddsp_amd as ddsp
rng = np.random.default_rng(1)
for B, N in ((9, 64000), (3, 12345), (128, 64000)):
  t = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N))); a = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N)))
  loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
  v = loss(t, a)
  ag = a.clone().requires_grad_(True); v2 = loss(t, ag); v2.backward()
  print(B, N, float(v).hex(), float(v2).hex(), float(ag.grad.double().abs().sum()))
PY
python /tmp/bits.py | tee $OUT/bits_xcd.txt
DDSP_EXP_SL_PLAIN_ORDER=1 python /tmp/bits.py | tee $OUT/bits_plain.txt
cmp $OUT/bits_xcd.txt $OUT/bits_plain.txt && echo "bits: identical (the gradient's checksum too)" || echo "bits: DIFFER (the gradient goes through fp32 atomics: compare the first two columns)"
echo "== FETCH_SIZE per launch, batch 128"
cd /tmp
for mode in xcd plain; do
  if [ $mode = plain ]; then export DDSP_EXP_SL_PLAIN_ORDER=1; else unset DDSP_EXP_SL_PLAIN_ORDER; fi
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$mode -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_loss_cmd.py 128 > $OUT/pmc_$mode.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$mode 2>/dev/null | grep -A3 "== stft_l1" | grep "==\|FETCH" | tee $OUT/fetch_$mode.txt
  rm -rf $OUT/pmc_$mode
done
echo "== done"
