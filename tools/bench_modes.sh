#!/bin/bash
# bench at B=32 and B=128 (the JSON carries both issue modes)
for B in 32 128; do
  echo "== batch $B"
  timeout 600 python bench.py --steps 50 --warmup 5 --batch $B --no-cpu-baseline --also-other-mode 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('  value %.0f Msamples/s  ms/step %.4f [%s] | other mode: %.4f ms/step | roofline %s frac %.3f avg_us %.1f' % (r['value'], r['ms_per_step'], r['config']['streams'], r['other_issue_mode']['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'], r['roofline']['avg_launch_us']), r['kernel_breakdown_us_isolated'])"
done
