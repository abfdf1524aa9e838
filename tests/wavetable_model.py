"""numpy model of harm_table_kernel (ddsp_amd/csrc/harmonic_table.hip): same index arithmetic
(half-step table grid, half table + halos, mirrored lookups, tap pairs) in fp32, used to pin the
method on CPU before the kernel is trusted.  Test infrastructure only."""
import os
import re
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COEFFS = os.path.join(os.path.dirname(HERE), 'ddsp_amd', 'csrc', 'wavetable_coeffs.h')
F32 = np.float32


def load_coeffs(path=COEFFS):
  txt = open(path).read()
  out = {}
  for m in re.finditer(r'float (\w+)\[(\d+)\] = \{(.*?)\};', txt, re.S):
    out[m.group(1)] = np.array([float(v.rstrip('f')) for v in m.group(3).replace('\n', ' ').split(',') if v.strip()],
                               dtype=F32)
  for m in re.finditer(r'(kWtDeg[EO]\d+) = (\d+)', txt):
    out[m.group(1)] = int(m.group(2))
  return out


def exp_sigmoid(x):
  x = x.astype(np.float64)
  return (2.0 * (1.0 / (1.0 + np.exp(-x))) ** np.log(10.0) + 1e-7).astype(F32)


def harmonic_table_model(amplitudes, hd, f0_hz, n_samples, sample_rate=16000, T=512, W=6, amp_linear=False):
  c = load_coeffs()
  B, Fr, K = hd.shape
  hop = n_samples // Fr
  H = 4
  NQ = T // 4
  de, do = c['kWtDegE%d' % W], c['kWtDegO%d' % W]
  ce = c['kWtE%d' % W].reshape(W // 2, de + 1)
  co = c['kWtO%d' % W].reshape(W // 2, do + 1)
  # 1 / psi_hat depends on k / T alone: the T = 512 table read at k 512 / T (csrc/harm_table_frags.h)
  invpsi512 = c['kWtInvPsi%d_T512' % W]
  invpsi = np.zeros(max(K + 1, 2), F32)
  kmax = min(K, (len(invpsi512) - 1) * T // 512)
  invpsi[1:kmax + 1] = invpsi512[(np.arange(1, kmax + 1) * (512 // T))]
  nyq = F32(sample_rate / 2.0)
  f0 = f0_hz[..., 0].astype(F32)
  # phase A: controls (exp_sigmoid, frame-rate Nyquist mask, normalise), amp * distribution
  a = exp_sigmoid(amplitudes[..., 0])
  x = exp_sigmoid(hd)
  kk = np.arange(1, K + 1, dtype=F32)
  x = np.where(f0[..., None] * kk >= nyq, F32(0), x)
  s = x.sum(-1, dtype=F32)
  x = x * (F32(1) / np.where(s == 0, F32(1e-7), s))[..., None]
  rows = (a[..., None] * x).astype(F32)                                  # [B,F,K]
  rows = np.concatenate([rows, rows[:, -1:]], axis=1)                    # halo row: frame F-1 held
  # table GEMM on the quarter range, odd and even harmonics apart
  n = np.arange(NQ)
  ang = (kk[None, :] * (2 * n[:, None] + 1)) / (2.0 * T)                 # revolutions, exact in fp32
  # round 4: 1 / psi_hat is part of the constant factor (the kernel's A-fragments), the rows are the plain amplitudes
  sinm = (np.sin(2 * np.pi * ang).astype(F32).astype(np.float64) * invpsi[1:K + 1].astype(np.float64)).astype(F32)      # [NQ,K]
  cdec = rows
  odd, even = np.arange(0, K, 2), np.arange(1, K, 2)                    # k = 1,3,.. / 2,4,..
  # both factors as two fp16 numbers, x = hi + lo / 2048; three products accumulated in fp32 (the kernel's MFMAs)
  def split(v):
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(F32)) * F32(2048)).astype(np.float16)
    return hi.astype(F32), lo.astype(F32)
  ah, al = split(sinm)
  bh, bl = split(cdec)
  def prod(sel):
    main = np.einsum('nk,bfk->bfn', ah[:, sel], bh[..., sel]).astype(F32)
    cross = (np.einsum('nk,bfk->bfn', ah[:, sel], bl[..., sel]) +
             np.einsum('nk,bfk->bfn', al[:, sel], bh[..., sel])).astype(F32)
    return (main + cross * F32(1.0 / 2048)).astype(F32)
  O, E = prod(odd), prod(even)
  half = T // 2
  tab = np.zeros((B, Fr + 1, half + 2 * H), F32)                        # index p + H, p = -H .. half+H-1
  tab[..., H + n] = O + E
  tab[..., H + (half - 1 - n)] = O - E
  m = np.arange(H)
  tab[..., H - 1 - m] = -tab[..., H + m]
  tab[..., H + half + m] = -tab[..., H + half - 1 - m]
  # phase: fp64 closed form per frame, per sample
  f64 = f0.astype(np.float64)
  csum = np.concatenate([np.zeros((B, 1)), np.cumsum(f64, axis=1)[:, :-1]], axis=1)
  run = hop * csum + (f64 - f64[:, :1]) * (hop - 1) / 2.0
  theta0 = (run / sample_rate) % 1.0
  fnext = np.concatenate([f64[:, 1:], f64[:, -1:]], axis=1)
  w = f64 / sample_rate
  dw = (fnext - f64) / sample_rate / (2.0 * hop)
  r = np.arange(hop, dtype=np.float64)
  cyc = theta0[..., None] + (r + 1.0) * (w[..., None] + dw[..., None] * r)       # [B,F,hop]
  th = (cyc - np.floor(cyc)).astype(F32)
  neg = th >= F32(0.5)
  th2 = np.where(neg, F32(1) - th, th)
  p = (th2 * F32(T) - F32(0.5)).astype(F32)        # fma in the kernel: single rounding; exact enough here
  fl = np.floor(p)
  z = ((p - fl) - F32(0.5)).astype(F32)
  i0 = fl.astype(np.int64)
  z2 = (z * z).astype(F32)
  acc0 = np.zeros(th.shape, F32); acc1 = np.zeros(th.shape, F32)
  bi, fi = np.meshgrid(np.arange(B), np.arange(Fr), indexing='ij')
  for pr in range(W // 2):
    e = np.zeros_like(z) + ce[pr][de]
    for d in range(de - 1, -1, -1):
      e = (e * z2 + ce[pr][d]).astype(F32)
    o = np.zeros_like(z) + co[pr][do]
    for d in range(do - 1, -1, -1):
      o = (o * z2 + co[pr][d]).astype(F32)
    # the tap at signed distance x = z + tau lies *below* the centre: index i0 - pr; x = z - tau: i0 + 1 + pr
    w_lo = (e + z * o).astype(F32)
    w_hi = (e - z * o).astype(F32)
    for wt, idx in ((w_lo, i0 - pr), (w_hi, i0 + 1 + pr)):
      t0 = tab[bi[..., None], fi[..., None], idx + H]
      t1 = tab[bi[..., None], fi[..., None] + 1, idx + H]
      acc0 = (acc0 + wt * t0).astype(F32)
      acc1 = (acc1 + wt * t1).astype(F32)
  lerp = (np.arange(hop, dtype=F32) / F32(hop)).astype(F32)
  w_next = lerp if amp_linear else (F32(0.5) - F32(0.5) * np.cos(np.pi * lerp)).astype(F32)
  w_cur = F32(1) - w_next
  out = w_cur * acc0 + w_next * acc1
  out = np.where(neg, -out, out)
  # audio-rate Nyquist mask (core.py:942-944): subtract what the table carries for masked samples
  fj, fj1 = f0, np.concatenate([f0[:, 1:], f0[:, -1:]], axis=1)
  for k in range(K):
    kf = F32(k + 1)
    top, bot = fj * kf, fj1 * kf
    cross = (np.maximum(top, bot) >= nyq * F32(1 - 4e-6)) & (np.minimum(top, bot) < nyq * F32(1 + 4e-6))
    if not cross.any():
      continue
    fk = (top[..., None] + ((bot - top)[..., None] * lerp).astype(F32)).astype(F32)
    sv = np.sin(2 * np.pi * ((th.astype(np.float64) * (k + 1)) % 1.0)).astype(F32)
    ak = w_cur * rows[:, :-1, k, None] + w_next * rows[:, 1:, k, None]
    out = out - np.where(cross[..., None] & (fk >= nyq), ak * sv, F32(0))
  return out.reshape(B, n_samples).astype(F32)
