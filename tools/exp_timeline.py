"""In-kernel phase timeline of harm_fused_kernel (debug flag 0x02000000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddsp_amd import _lib, core, build
build.build()
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F, K, N, sr = 1000, 100, 64000, 16000
rng = np.random.default_rng(0)
amps = core.tf_float32(rng.standard_normal((B, F, 1)))
hd = core.tf_float32(rng.standard_normal((B, F, K)))
f0 = core.tf_float32(70 + rng.standard_normal((B, F, 1)))
audio = torch.empty((B, N), device='cuda')
ws = torch.empty(lib.ddsp_harmonic_workspace_bytes(B, F, K, N), dtype=torch.uint8, device='cuda')
dbg = torch.zeros((2048, 16), dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def call(flags, dbgptr):
  lib.ddsp_harmonic_f32(amps.data_ptr(), hd.data_ptr(), f0.data_ptr(), audio.data_ptr(), dbgptr, None,
                        ws.data_ptr(), ws.numel(), B, F, K, N, sr, flags, st)
for _ in range(3): call(0x3, None)
torch.cuda.synchronize()
call(0x3 | 0x02000000, dbg.data_ptr())
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
nb = min(2048, B * ((F + 15) // 16))
d = d[:nb]
t0 = d[:, 0].min()
d = np.where(d > 0, (d - t0) * 0.01, np.nan)     # 100 MHz -> us
names0 = ['start', 'barrier 1 (loads issued, f0 in LDS)', 'values scaled in LDS', 'row sums', 'stores issued', 'tables + stores acked', 'tables (dup)', 'tiles done']
names = names0 + ['u2: ' + n for n in names0[1:]] + ['u3: ' + n for n in names0[1:]]
print('B=%d  blocks=%d  (us since first block start; min / median / max over blocks)' % (B, nb))
for i in range(16):
  col = d[:, i]
  if np.all(np.isnan(col)): break
  nm = names[i] if i < len(names) else '...'
  print('  %-34s %7.2f %7.2f %7.2f   pct 10/25/75/90: %s  n=%d' % (nm, np.nanmin(col), np.nanmedian(col), np.nanmax(col), ' '.join('%.1f' % np.nanpercentile(col, q) for q in (10, 25, 75, 90)), np.sum(~np.isnan(col))))
# ---- where are the stragglers? ----
arr = d[:, 5]            # phase A done
blk = np.arange(nb)
print('loads-arrived by XCD (blockIdx % 8):', ' '.join('%.1f' % np.nanmedian(arr[blk % 8 == x]) for x in range(8)))
print('loads-arrived max by XCD           :', ' '.join('%.1f' % np.nanmax(arr[blk % 8 == x]) for x in range(8)))
c = blk % 63
print('loads-arrived median by frame-chunk c (0,10,..60):', ' '.join('%.1f' % np.nanmedian(arr[c == x]) for x in range(0, 63, 10)))
order = np.argsort(arr)
print('slowest 12 blocks:', [(int(i), round(float(arr[i]), 1)) for i in order[-12:]])
print('fastest 6 blocks :', [(int(i), round(float(arr[i]), 1)) for i in order[:6]])
print('percentiles of loads-arrived: ', ' '.join('%d%%:%.1f' % (q, np.nanpercentile(arr, q)) for q in (10, 25, 50, 75, 90, 95, 99)))
st = d[:, 0]
print('start time percentiles        : ', ' '.join('%d%%:%.1f' % (q, np.nanpercentile(st, q)) for q in (50, 90, 99, 100)))
cu_slot = blk // 8 % 32
print('loads-arrived median by (blockIdx//8)%32 [CU within XCD?] first 8:', ' '.join('%.1f' % np.nanmedian(arr[cu_slot == x]) for x in range(8)))
late = (blk >= 1792)
print('median arrival blocks <1792: %.1f   blocks >=1792: %.1f' % (np.nanmedian(arr[~late]), np.nanmedian(arr[late])))
