#!/bin/bash
# bench in both stream modes at B=32 and B=128
for B in 32 128; do
  for M in "" "--overlap"; do
    echo "== batch $B $M"
    timeout 600 python bench.py --steps 50 --warmup 5 --batch $B --no-cpu-baseline $M 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('  value %.0f Msamples/s  ms/step %.4f  roofline %s frac %.3f avg_us %.1f' % (r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['avg_launch_us']), r['kernel_breakdown_us'])"
  done
done
