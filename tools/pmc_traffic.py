"""Turn rocprofv3 FETCH_SIZE / WRITE_SIZE passes into HBM bytes per kernel launch.

Units and gfx950 correction (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read on gfx950,
so read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken at face value (uncalibrated).
"""
import collections, csv, glob, json, os, sys

root, batch = sys.argv[1], int(sys.argv[2])
# optional: n_frames n_harmonics n_samples sample_rate of the launch (bench.py matches a record on kernel, batch AND shape)
shape = [int(v) for v in sys.argv[3:7]]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
  with open(path) as f:
    for row in csv.DictReader(f):
      name = row.get('Kernel_Name', '')
      if 'ddsp::' not in name:
        continue
      short = name.split('(')[0].replace('void ', '').replace('ddsp::', '').split('<')[0]
      vals[short][row['Counter_Name']].append(float(row['Counter_Value']))
alias = {'tv_fir128_kernel': 'tv_fir_kernel', 'noise_ir65_kernel': 'noise_ir_kernel'}
out = {'batch': batch, 'shape': dict(zip(('n_frames', 'n_harmonics', 'n_samples', 'sample_rate'), shape)) if len(shape) == 4 else {},
       'note': 'bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 '
       '(gfx950 FETCH_SIZE correction x2; WRITE_SIZE uncalibrated)', 'kernels': {}, 'detail': {}}
for k, d in vals.items():
  fetch = sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1)
  write = sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1)
  total = (2.0 * fetch + write) * 1024.0
  out['kernels'][alias.get(k, k)] = total
  out['detail'][k] = {'FETCH_SIZE_KiB': fetch, 'WRITE_SIZE_KiB': write,
                      'read_bytes_corrected': 2.0 * fetch * 1024.0, 'write_bytes': write * 1024.0,
                      'launches': len(d['FETCH_SIZE'])}
print(json.dumps(out, indent=1))
