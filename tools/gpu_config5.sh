#!/bin/bash
# BASELINE configs[4] per GPU (48 kHz, 200 harmonics, 10 s clips, frame size 192, batch 32): the wavetable kernel's WIDE
# instance against the direct sum, one stream and two, three f0 regimes.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_config5.sh [tag]'
TAG=${1:-r03u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (the shapes this kernel path takes)"
timeout 600 python -m pytest tests -m gpu -q -k "129_to_200 or config5 or not_multiples_of_four or fused_add or harmonic_random_shapes" 2>&1 | tee $OUT/pytest_gpu_wide.txt | tail -4
C5="--batch 32 --n-frames 2500 --n-harmonics 200 --n-samples 480000 --sample-rate 48000 --steps 200 --warmup 60 --no-cpu-baseline --no-aux --no-second-shape"
for f0 in 70 100 440; do
  for hk in auto direct; do
    for st in 1 2; do
      timeout 300 python bench.py $C5 --f0 $f0 --harm-kernel $hk --streams $st 2>/dev/null | tail -1 > $OUT/bench_c5_f0${f0}_${hk}_s${st}.json
      python - <<PY
import json
d = json.load(open('$OUT/bench_c5_f0${f0}_${hk}_s${st}.json'))
print('f0 $f0 $hk streams $st: %.1f us/step  %.0f Msamples/s  isolated %s' % (d['ms_per_step'] * 1e3, d['value'], d.get('kernel_breakdown_us_isolated')))
PY
    done
  done
done
echo "== the bench headline (batch 128, K = 100), unchanged?"
timeout 300 python bench.py --no-cpu-baseline --no-aux --no-second-shape 2>/dev/null | tail -1 > $OUT/bench_b128.json
python -c "
import json; d = json.load(open('$OUT/bench_b128.json')); print('B=128: %.2f us/step' % (d['ms_per_step'] * 1e3), d.get('kernel_breakdown_us_isolated'))"
echo "== done"
