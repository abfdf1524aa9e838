"""fp32 error of the Chebyshev-blocked sine bank used by harm_synth_kernel (numpy emulation).

Two exact seeds per block (fractional phase by fma(k, theta, -rint(k theta)), then sin) and
s_{k+1} = 2cos(2 pi theta) s_k - s_{k-1}.  Prints the max error per harmonic over random and
adversarial (theta near 0, 1/2, 1/4) phases.  python tools/cheb_error.py
"""
import numpy as np

f32 = np.float32
rng = np.random.default_rng(0)
theta = np.concatenate([rng.uniform(0, 1, 200000), rng.uniform(0, 2e-3, 50000),
                        0.5 + rng.uniform(-2e-3, 2e-3, 50000),
                        0.25 + rng.uniform(-2e-3, 2e-3, 20000)]).astype(f32)
K = 100
th64 = theta.astype(np.float64)
exact = np.sin(2 * np.pi * np.outer(th64, np.arange(1, K + 1)))


def seed(k):
  p = th64 * k
  return np.sin(2 * np.pi * (p - np.rint(p)).astype(f32).astype(np.float64)).astype(f32)


def fma32(a, b, c):
  return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


for block in (8, 16, 32):
  out = np.zeros((len(theta), K), f32)
  c2 = (2 * np.cos(2 * np.pi * th64)).astype(f32)
  for k0 in range(1, K + 1, block):
    ks = list(range(k0, min(k0 + block, K + 1)))
    s0 = seed(ks[0]); out[:, ks[0] - 1] = s0
    if len(ks) > 1:
      s1 = seed(ks[1]); out[:, ks[1] - 1] = s1
      for k in ks[2:]:
        s2 = fma32(c2, s1, -s0); out[:, k - 1] = s2; s0, s1 = s1, s2
  print('block %2d: max |err| per harmonic %.2e' % (block, np.abs(out - exact).max()))
direct = np.sin(2 * np.pi * np.outer(theta, np.arange(1, K + 1, dtype=f32)).astype(f32).astype(np.float64))
print('direct sin(fl32(k*theta)): max |err| %.2e' % np.abs(direct - exact).max())

# ---- stride-2 variant used by harm_fused_kernel: s[h] = 2cos(4 pi theta) s[h-2] - s[h-4], four exact
# seeds per super-block of SB harmonics (two independent chains: odd and even harmonics) ------------
for sb in (16, 32, 64):
  out = np.zeros((len(theta), K), f32)
  c4 = (2 * np.cos(4 * np.pi * th64)).astype(f32)
  for k0 in range(1, K + 1, sb):
    ks = list(range(k0, min(k0 + sb, K + 1)))
    hist = []
    for i, k in enumerate(ks):
      if i < 4:
        v = seed(k)
      else:
        v = fma32(c4, hist[-2], -hist[-4])
      out[:, k - 1] = v
      hist.append(v)
  print('stride-2, super-block %2d: max |err| per harmonic %.2e' % (sb, np.abs(out - exact).max()))

# ---- what harm_fused_kernel does now: two exact seeds, the next two by two steps of the stride-1
# recurrence, then stride 2; for K = 100 the runs are harmonics 1..64 and 65..100 -------------------
out = np.zeros((len(theta), K), f32)
c2 = (2 * np.cos(2 * np.pi * th64)).astype(f32)
c4 = (2 * np.cos(4 * np.pi * th64)).astype(f32)
for k0, k1 in ((1, 64), (65, 100)):
  hist = []
  for i, k in enumerate(range(k0, k1 + 1)):
    if i < 2:
      v = seed(k)
    elif i < 4:
      v = fma32(c2, hist[-1], -hist[-2])
    else:
      v = fma32(c4, hist[-2], -hist[-4])
    out[:, k - 1] = v
    hist.append(v)
print('fused kernel (2 exact + 2 derived seeds, runs 1-64 / 65-100): max |err| per harmonic %.2e' %
      np.abs(out - exact).max())
