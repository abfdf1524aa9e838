"""Per-phase clock stamps of block 0 of harm_bwd_table_kernel (DDSP_EXP_BT_TIMELINE): for every group of frames and wavefront the
clocks spent in: spreading | fold | wait at barrier 1 | product + masks | wait at barrier 2 | stores.

    python tools/exp_bwd_timeline.py [batch]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(16 * 8 * 8, dtype=torch.int64, device='cuda')
os.environ['DDSP_EXP_BT_TIMELINE'] = str(buf.data_ptr())
import ddsp_amd as ddsp
T = ddsp.core.tf_float32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F, K, N = 1000, 100, 64000
rng = np.random.default_rng(0)
amps = T(rng.standard_normal((B, F, 1))).requires_grad_(True)
hd = T(rng.standard_normal((B, F, K))).requires_grad_(True)
f0 = T(70.0 + rng.standard_normal((B, F, 1)))
g = T(rng.standard_normal((B, N)))
synth = ddsp.synths.Harmonic(n_samples=N)
for _ in range(5):
  amps.grad = None; hd.grad = None
  synth(amps, hd, f0).backward(g)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(16, 8, 8)
t0 = t[0, :, 0].min()
names = ['spread', 'fold', 'wait1', 'product', 'wait2', 'store']
print('s_memtime clocks; per group: its start and length in clocks since the block\'s first stamp, then per wavefront the clocks spent in each phase')
for it in range(16):
  if t[it, 0, 6] == 0: break
  print('group %2d  start %7d   length %6d' % (it, t[it, :, 0].min() - t0, t[it, :, 6].max() - t[it, :, 0].min()))
  for w in range(8):
    d = [int(t[it, w, k + 1] - t[it, w, k]) for k in range(6)]
    print('    wave %d  ' % w + '  '.join('%s %5d' % (n, v) for n, v in zip(names, d)))
