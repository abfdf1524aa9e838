#!/bin/bash
# SpectralLoss forward: where the time goes - the kernel with its load phase / transforms / per-bin part compiled out
# (-DDDSP_SL_NO_LOAD / NO_FFT / NO_BINS variants from tools/build_variant.sh), per FFT size and all six, one gpurun call.
# Usage: gpurun -- 'bash tools/exp_loss_ablation.sh <tag> <batch> lib1.so lib2.so ...'
TAG=$1; B=$2; shift; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
cat > /tmp/loss_sizes.py <<'PY'
import json, os, sys, time
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib
B = int(sys.argv[1]); N = 64000
rng = np.random.default_rng(1)
a = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N))); t = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N)))
res = {}
for sizes in ((2048,), (1024,), (512,), (256,), (128,), (64,), (2048, 1024, 512, 256, 128, 64)):
  loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=1.0, logmag_weight=1.0)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.05:
    for _ in range(5): loss(t, a)
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=256)
  for _ in range(20): loss(t, a)
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res['all' if len(sizes) > 1 else str(sizes[0])] = round(sum(v[0] for v in bd.values()) / 20 * 1e3, 1)
ag = a.clone().requires_grad_(True)
loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
for _ in range(5):
  ag.grad = None; loss(t, ag).backward()
_lib.profile_begin(None, max_records=256)
for _ in range(10):
  ag.grad = None; loss(t, ag).backward()
torch.cuda.synchronize()
bd = _lib.profile_end()
res['value_and_grad'] = {k: round(v[0] / v[1] * 1e3, 1) for k, v in bd.items()}
print(json.dumps(res))
PY
for LIB in ddsp_amd/lib/libddsp_amd.so "$@"; do
  echo "$(basename $LIB .so) B=$B: $(timeout 300 python tools/with_lib.py $LIB /tmp/loss_sizes.py $B 2>/dev/null | tail -1)"
done 2>&1 | tee $OUT/loss_ablation_b$B.txt
