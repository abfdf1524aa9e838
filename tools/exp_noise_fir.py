"""FilteredNoise at the bench shape: the matrix-core kernel (noise_mfma65_kernel: IR design + FIR as fp16 hi/lo-split MFMA
products) against noise_fused65_kernel (FIR on the vector ALUs) with its IR designs; per-launch time from dispatch events,
agreement between them, and the in-kernel phase timeline of the matrix-core kernel.

    python tools/exp_noise_fir.py [batch ...]

DDSP_MF_DBG_WAVE=8..15 picks the FIR wavefront (and producer wavefront - 8) whose per-tick stamps the timeline shows: the
youngest wavefront of a SIMD is the slow one (profiles/r02n_noise_mfma_v7_per_wavefront_timeline.txt).  The FIR wavefronts of
the default build (two blocks of 8 wavefronts per CU) are 8..11; DDSP_MF_DBG_BLOCK=<block> picks the block (blocks >= the CU
count are the second, younger blocks of their CUs).
"""
import json, os, sys, time
os.environ.setdefault('DDSP_NOISE_DEBUG_TIMELINE', '1')      # lets flag bit 30 through (the stamp buffer of the timeline below)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, core, build
build.build()
lib = _lib.load()
F, M, N = 1000, 65, 64000
for B in [int(v) for v in sys.argv[1:]] or [32, 128]:
  rng = np.random.default_rng(0)
  mags = core.tf_float32(rng.standard_normal((B, F, M)))
  res, outs = {'batch': B}, {}
  for name, kernel in (('mfma', 'auto'), ('vector_fir', 'vector')):
    synth = ddsp.synths.FilteredNoise(n_samples=N, window_size=0, seed=7)
    synth.kernel = kernel
    for _ in range(20): synth(mags)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.05:
      for _ in range(20): synth(mags)
      torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=512)
    for _ in range(50): synth(mags)
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    steps = 300
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): synth(mags)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    synth._calls = 0
    outs[name] = synth(mags)
    res[name] = {'us_per_call_back_to_back': dt * 1e6, 'kernel_us': {k: v[0] / v[1] * 1e3 for k, v in bd.items()}}
  res['max_abs_diff'] = float((outs['mfma'] - outs['vector_fir']).abs().max())
  res['max_abs_out'] = float(outs['mfma'].abs().max())
  print(json.dumps(res))
  # per-tick timeline of block 0 of the persistent matrix-core kernel (debug flag 0x40000000: the controls pointer
  # carries the stamp buffer [tick + 1][role: design, noise, FIR][begin, end], 100 MHz wall clock)
  audio = torch.empty((B, N), device='cuda')
  ws = torch.empty(max(lib.ddsp_filtered_noise_workspace_bytes(B, F, M, N, 0), 16), dtype=torch.uint8, device='cuda')
  dbg = torch.zeros((16, 3, 2), dtype=torch.int64, device='cuda')
  rc = lib.ddsp_filtered_noise_f32(mags.data_ptr(), None, audio.data_ptr(), dbg.data_ptr(), ws.data_ptr(), ws.numel(),
                                   B, F, M, N, 0, -5.0, 1 | 0x40000000, 1, 0, torch.cuda.current_stream().cuda_stream)
  assert rc == 0, rc
  torch.cuda.synchronize()
  d = dbg.cpu().numpy().astype(np.float64)
  t0 = d[0, 0, 0]
  print('noise_mfma65_kernel B=%d, block 0: per tick (us since the block started) role begin..end' % B)
  for k in range(16):
    if d[k].max() == 0: break
    row = ['tick %2d' % (k - 1)]
    for r, nm in enumerate(('producer', 'its design part', 'FIR')):
      if d[k, r, 0] > 0: row.append('%s %6.2f..%6.2f (%.2f)' % (nm, (d[k, r, 0] - t0) * 0.01, (d[k, r, 1] - t0) * 0.01, (d[k, r, 1] - d[k, r, 0]) * 0.01))
    print('  ' + '   '.join(row))
