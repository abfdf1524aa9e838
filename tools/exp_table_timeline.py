"""Per-tick timeline of harm_table_kernel's block 0 (shader clocks): DDSP_EXP_TABLE_TIMELINE=1 makes the launch
record and print it.   python tools/exp_table_timeline.py [batch]"""
import os, sys
os.environ['DDSP_EXP_TABLE_TIMELINE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F, K, N = 1000, 100, 64000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
synth = ddsp.synths.Harmonic(n_samples=N)
for i in range(3):
  sys.stderr.write('--- launch %d\n' % i)
  synth(amps, hd, f0)
torch.cuda.synchronize()
