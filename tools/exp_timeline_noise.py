"""In-kernel phase timeline of noise_fused65_kernel (debug flag 0x40000000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddsp_amd import _lib, core, build
build.build()
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
EXTRA = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0     # 0x2: IR design on the matrix cores, 0x6: ... from registers
F, M, N = 1000, 65, 64000
rng = np.random.default_rng(0)
mags = core.tf_float32(rng.standard_normal((B, F, M)))
audio = torch.empty((B, N), device='cuda')
ws = torch.empty(lib.ddsp_filtered_noise_workspace_bytes(B, F, M, N, 0), dtype=torch.uint8, device='cuda')
nblk = B * 17
dbg = torch.zeros((nblk, 8), dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def call(flags, dbgptr):
  rc = lib.ddsp_filtered_noise_f32(mags.data_ptr(), None, audio.data_ptr(), dbgptr, ws.data_ptr(), ws.numel(),
                                   B, F, M, N, 0, -5.0, flags, 1, 0, st)
  assert rc == 0, rc
for _ in range(3): call(1 | EXTRA, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call(1 | EXTRA, None)
e1.record(); torch.cuda.synchronize()
print('B=%d flags+0x%x: %.1f us per call (events, incl. launch gaps)' % (B, EXTRA, e0.elapsed_time(e1) / 10 * 1e3))
call(1 | EXTRA | 0x40000000, dbg.data_ptr())
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
t0 = d[:, 0].min()
d = (d - t0) * 0.01
names = ['start', 'magnitudes staged', 'IR designed', 'noise tile staged', 'FIR done', 'stored']
print('blocks=%d  (us since first block start; min / median / max over blocks)' % nblk)
for i, nm in enumerate(names):
  print('  %-22s %7.2f %7.2f %7.2f' % (nm, d[:, i].min(), np.median(d[:, i]), d[:, i].max()))
