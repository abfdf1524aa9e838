"""Average rocprofv3 --pmc counter values per kernel over all dispatches in a directory tree."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
  with open(path) as f:
    for row in csv.DictReader(f):
      name = row.get('Kernel_Name', '')
      short = name.split('(')[0].replace('void ', '').replace('ddsp::', '')
      if not any(k in short for k in ('harm_', 'noise_', 'tv_fir', 'add_', 'uniform_')):
        continue
      acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
for kern in sorted(acc):
  print('==', kern)
  for c in sorted(acc[kern]):
    v = acc[kern][c]
    print('   %-26s avg %16.1f   (n=%d)' % (c, sum(v) / len(v), len(v)))
