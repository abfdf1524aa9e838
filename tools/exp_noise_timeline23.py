"""Per-tick timeline of noise_mfma65_kernel with 23-bit generated noise (the default since round 6): one process per stamped
wavefront pair (DDSP_MF_DBG_WAVE = 8 .. 11: producer wavefront w - 8 - designers 0, 1, noise makers 2, 3 - and FIR wavefront w - 4).

    python tools/exp_noise_timeline23.py [batch] [bits]
"""
import json, os, subprocess, sys
CHILD = r'''
import os, sys
os.environ.setdefault('DDSP_NOISE_DEBUG_TIMELINE', '1')
sys.path.insert(0, %(root)r)
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, core
lib = _lib.load()
F, M, N, B, bits = 1000, 65, 64000, %(batch)d, %(bits)d
rng = np.random.default_rng(0)
mags = core.tf_float32(rng.standard_normal((B, F, M)))
audio = torch.empty((B, N), device='cuda')
ws = torch.empty(max(lib.ddsp_filtered_noise_workspace_bytes(B, F, M, N, 0), 16), dtype=torch.uint8, device='cuda')
flags = 1 | 0x40000000 | (0x10 if bits == 23 else 0)
for it in range(3):
  dbg = torch.zeros((16, 3, 2), dtype=torch.int64, device='cuda')
  rc = lib.ddsp_filtered_noise_f32(mags.data_ptr(), None, audio.data_ptr(), dbg.data_ptr(), ws.data_ptr(), ws.numel(),
                                   B, F, M, N, 0, -5.0, flags, 1, 0, torch.cuda.current_stream().cuda_stream)
  assert rc == 0, rc
  torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
t0 = d[0, 0, 0] if d[0, 0, 0] > 0 else d[d > 0].min()
for k in range(16):
  if d[k].max() == 0: break
  row = ['tick %%2d' %% (k - 1)]
  for r, nm in enumerate(('producer', 'its design part', 'FIR')):
    if d[k, r, 0] > 0: row.append('%%s %%6.2f..%%6.2f (%%.2f)' %% (nm, (d[k, r, 0] - t0) * 0.01, (d[k, r, 1] - t0) * 0.01, (d[k, r, 1] - d[k, r, 0]) * 0.01))
  print('  ' + '   '.join(row))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 23
for w in (8, 9, 10, 11):
  env = dict(os.environ, DDSP_MF_DBG_WAVE=str(w))
  out = subprocess.run([sys.executable, '-c', CHILD % {'root': root, 'batch': batch, 'bits': bits}], capture_output=True, text=True, env=env)
  print('== DDSP_MF_DBG_WAVE=%d (producer wavefront %d = %s, FIR wavefront %d), batch %d, %d-bit noise' % (w, w - 8, 'designer' if w < 10 else 'noise maker', w - 4, batch, bits))
  print(out.stdout[-3000:] if out.returncode == 0 else out.stderr[-1500:])
