import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib
T=ddsp.core.tf_float32
res={}
for name,B,F,K,N,sr,f0c in [('config5_b8', 8, 2500, 200, 480000, 48000, 110.0), ('k160_b32', 32, 1000, 160, 64000, 32000, 70.0)]:
  rng=np.random.default_rng(0)
  amps=T(rng.standard_normal((B,F,1))).requires_grad_(True); hd=T(rng.standard_normal((B,F,K))).requires_grad_(True)
  f0=T(f0c+rng.standard_normal((B,F,1))); g=T(rng.standard_normal((B,N)))
  synth=ddsp.synths.Harmonic(n_samples=N, sample_rate=sr)
  def fn():
    amps.grad=None; hd.grad=None
    synth(amps,hd,f0).backward(g)
  for _ in range(3): fn()
  torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=256)
  for _ in range(10): fn()
  torch.cuda.synchronize()
  bd=_lib.profile_end()
  res[name]={k.replace('_kernel','').replace('harm_',''): round(v[0]/v[1]*1e3,1) for k,v in bd.items()}
print(json.dumps(res))
