#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for B in 32 128; do for FL in 0x0 0x20000000 0x28000000; do
  rm -rf /tmp/exp_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/exp_prof -o t -- python $GRAFT_REPO_ROOT/tools/exp_harm.py $B $FL > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open('/tmp/exp_prof/t_kernel_stats.csv')):
    if 'harm_fused' in r['Name']: print('B=$B flags=$FL  calls', r['Calls'], 'avg_us %.1f min_us %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done; done
