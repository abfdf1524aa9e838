"""Run the fused harmonic kernel N times with given debug flag bits (for rocprofv3 --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddsp_amd import _lib, core, build
build.build()
lib = _lib.load()
B = int(sys.argv[1]); extra = int(sys.argv[2], 0)
F, K, N, sr = 1000, 100, 64000, 16000
rng = np.random.default_rng(0)
amps = core.tf_float32(rng.standard_normal((B, F, 1)))
hd = core.tf_float32(rng.standard_normal((B, F, K)))
f0 = core.tf_float32(70 + rng.standard_normal((B, F, 1)))
audio = torch.empty((B, N), device='cuda')
ws = torch.empty(lib.ddsp_harmonic_workspace_bytes(B, F, K, N), dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
  lib.ddsp_harmonic_f32(amps.data_ptr(), hd.data_ptr(), f0.data_ptr(), audio.data_ptr(), None, None,
                        ws.data_ptr(), ws.numel(), B, F, K, N, sr, 0x3 | extra, st)
  torch.cuda.synchronize()
