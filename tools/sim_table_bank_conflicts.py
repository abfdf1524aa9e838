"""LDS bank conflicts of harm_table_kernel's table reads as a function of f0 (a model, CPU only).

Phase B reads, per sample, six neighbouring entries of two table rows (W = 6: floor(pos) - 2 .. + 3, as three pairs) at
pos = 512 fold(theta) - 1/2; a wavefront instruction is serviced in two groups of 32 lanes, one LDS cycle per group when
no two lanes of the group want DIFFERENT addresses on the same bank (bank = dword address mod 32 for 4-byte reads; the same
address is a broadcast).  Lanes = 64 consecutive samples of a frame: an arithmetic progression of positions with stride
512 f0 / sr, folded at 0 and 256.  The model counts, per read instruction, the largest number of distinct addresses on one
bank in each lane group, summed over the instructions of a tile and divided by the conflict-free count - the factor by which
the table reads' LDS time grows.  (What it costs the kernel depends on how much of the LDS's time those reads take: measured,
333 Hz held constant - factor 7.2 here against 2.0 at 70 Hz - runs at 58 us against 37 us, 338 Hz (3.5) at 40 us; profiles/r03w_f0_dependence_two_effects.txt.)

    python tools/sim_table_bank_conflicts.py [sample_rate]
"""
import sys
import numpy as np

SR = float(sys.argv[1]) if len(sys.argv) > 1 else 16000.0
T, H, TS, W = 512, 4, 268, 6


def tile_factor(f0, theta0, rng):
  r = np.arange(64)
  theta = (theta0 + (r + 1) * f0 / SR) % 1.0
  th = 0.5 - np.abs(0.5 - theta)
  pos = th * T - 0.5
  idx = np.floor(pos).astype(int)                     # -1 .. 255
  row = rng.integers(0, 31)
  total, ideal = 0, 0
  for rr in (row, row + 1):                           # the two table rows of the frame
    base = rr * TS + H + idx
    for pair in ((-2, -1), (0, 1), (2, 3)):           # one ds_read2_b32 (two dwords) per pair
      for off in pair:
        addr = base + off
        for g in (slice(0, 32), slice(32, 64)):
          a = addr[g]
          worst = max(len(set(a[a % 32 == b])) for b in range(32))
          total += worst
          ideal += 1
  return total / ideal


def main():
  rng = np.random.default_rng(0)
  f0s = np.arange(40.0, 1201.0, 1.0)
  fac = np.array([np.mean([tile_factor(f, rng.random(), rng) for _ in range(24)]) for f in f0s])
  base = np.median(fac)
  print('sample rate %g Hz, table of %d points: conflict factor of the table reads (1 = none), f0 = 40 .. 1200 Hz in 1 Hz steps' % (SR, T))
  print('median %.2f, minimum %.2f (f0 = %g), maximum %.2f (f0 = %g)' % (base, fac.min(), f0s[fac.argmin()], fac.max(), f0s[fac.argmax()]))
  for lo, hi in ((40, 100), (100, 200), (200, 400), (400, 800), (800, 1200)):
    m = (f0s >= lo) & (f0s < hi)
    print('  %4d .. %4d Hz: mean %.2f, share of f0 values with a factor above 1.5 x the median: %.0f %%' % (lo, hi, fac[m].mean(), 100 * (fac[m] > 1.5 * base).mean()))
  print('bands (factor > 1.5 x the median), as f0 ranges and the stride 512 f0 / sr at their peak:')
  above = fac > 1.5 * base
  i = 0
  while i < len(f0s):
    if above[i]:
      j = i
      while j + 1 < len(f0s) and above[j + 1]:
        j += 1
      k = i + int(np.argmax(fac[i:j + 1]))
      print('  %6.0f .. %6.0f Hz  peak %.2f at %g Hz (stride %.3f = 32 / %.3f)' % (f0s[i], f0s[j], fac[k], f0s[k], T * f0s[k] / SR, 32 / (T * f0s[k] / SR)))
      i = j + 1
    else:
      i += 1
  for f in (70, 128, 200, 203, 250, 254, 333, 335, 338, 345, 400, 407, 500, 508, 666, 677, 1000, 1015):
    print('  f0 %5d Hz: factor %.2f' % (f, fac[int(f - 40)]))


if __name__ == '__main__':
  main()
