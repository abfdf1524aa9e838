"""Average rocprofv3 --pmc counter values per kernel over all dispatches in a directory tree.

    python tools/pmc_summary.py <dir> [--json out.json batch [n_frames n_harmonics n_samples sample_rate]]

The JSON form (profiles/pmc_sq_*.json) is what bench.py reads for roofline.alu_note.executed."""
import collections
import csv
import glob
import os
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
  with open(path) as f:
    for row in csv.DictReader(f):
      name = row.get('Kernel_Name', '')
      short = name.split('(')[0].replace('void ', '').replace('ddsp::', '')
      if not any(k in short for k in ('harm_', 'noise_', 'tv_fir', 'add_', 'uniform_', 'stft_', 'rv_', 'spec_')):
        continue
      acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
for kern in sorted(acc):
  print('==', kern)
  for c in sorted(acc[kern]):
    v = acc[kern][c]
    print('   %-26s avg %16.1f   (n=%d)' % (c, sum(v) / len(v), len(v)))

if '--json' in sys.argv:
  i = sys.argv.index('--json')
  rest = [int(v) for v in sys.argv[i + 2:i + 7]]
  rec = {'batch': rest[0], 'shape': dict(zip(('n_frames', 'n_harmonics', 'n_samples', 'sample_rate'), rest[1:5])) if len(rest) == 5 else {},
         'source': 'rocprofv3 --pmc passes of tools/pmc.sh, averaged per launch (tools/pmc_summary.py)',
         'sq': {kern.split('<')[0]: {c: sum(v) / len(v) for c, v in acc[kern].items()} for kern in acc}}
  with open(sys.argv[i + 1], 'w') as f:
    json.dump(rec, f, indent=1, sort_keys=True)
