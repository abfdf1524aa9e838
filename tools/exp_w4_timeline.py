"""Per-unit timeline of one block of harm_wt4_kernel (shader clocks per wavefront and stage), from a -DDDSP_W4_TIMELINE build
(bash tools/build_variant.sh w4tl "harmonic_table.hip" -DDDSP_W4_TIMELINE).

    python tools/exp_w4_timeline.py [batch] [f0] [block]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['DDSP_EXP_TABLE_TIMELINE'] = '1'
if len(sys.argv) > 3: os.environ['DDSP_W4_DBG_BLOCK'] = sys.argv[3]
import numpy as np, torch
from ddsp_amd import _lib
_lib.LIB_PATH = os.environ.get('DDSP_TIMELINE_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libddsp_amd_w4tl.so')
import ddsp_amd as ddsp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
f0c = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
F, K, N, SR = 1000, 100, 64000, 16000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
f0 = ddsp.core.tf_float32(f0c + rng.standard_normal((B, F, 1)))
synth = ddsp.synths.Harmonic(n_samples=N, sample_rate=SR)
devnull = os.open(os.devnull, os.O_WRONLY)
saved = os.dup(2)
os.dup2(devnull, 2)                      # every launch prints a timeline: keep the last one only
for _ in range(5): synth(amps, hd, f0)
torch.cuda.synchronize()
os.dup2(saved, 2)
synth(amps, hd, f0)
torch.cuda.synchronize()
