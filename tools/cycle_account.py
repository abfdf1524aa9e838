"""Per-class cycle account of harm_table_kernel: instructions per tick and role, counted in the
gfx950 instruction stream hipcc emits for the default instantiation (K = 100, hop 64), priced with the issue costs
measured on the MI355X (profiles/r03a_microbench_issue_cost_by_class.txt, r03b_microbench_issue_cost_second_pass.txt:
clocks per instruction AS ONE WAVEFRONT SEES IT with four wavefronts per SIMD - the kernel's occupancy), against the
measured length of that role's part of a tick (profiles/r03q_timeline_final.txt).

A wavefront is in-order: the length of its tick is the sum of what its instructions cost it.  The tick of the block is
its slowest wavefront's; the SIMD-side view (clocks of SIMD time = wave price / 4) says how full the issue ports are.

    python tools/cycle_account.py            # CPU only (hipcc cross-compiles); prints the table
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# clocks per instruction as one wavefront sees it, W = 4 wavefronts per SIMD all issuing that class (microbench7 / 9)
PRICE = {
    'simple': 8.6,        # v_fma/add/mul/xor/and/or/mov/sub/max f32 and 32-bit integer, v_cndmask e64: 8.2 .. 9.4
    'class2': 12.5,       # cvt, floor/fract, lshl_add, fp64 (fma/add/fract/cvt/cmp), pk_fma/mul/add f32, mul_lo/hi, mad_u64, med3, bfi: 11.6 .. 14.2
    'trans': 13.7,        # v_exp/log/rcp/sin/cos, v_permlane16/32_swap: the quarter-rate unit is the SIMD's, and only ONE of a
                          # SIMD's four wavefronts (its row maker) issues these here: the W = 2 price (21.9 when all four do)
    'dpp': 12.5,          # v_*_dpp
    'readlane': 13.5,     # v_readlane / v_readfirstlane
    'salu': 14.5,         # s_add/mov/mul/cselect/and/or/lshl/cmp: 14.5 .. 15.2 alone; ~9.5 each when interleaved with vector work
    'swait': 6.9,         # s_waitcnt / s_nop (nothing pending)
    'branch': 9.0,        # s_cbranch / s_branch (taken or not; not measured separately: priced as an interleaved scalar op)
    'lds': 13.0,          # ds_read2_b32 / ds_read_b64 / ds_read_b128 / ds_write_b32 issued between vector work (81.9 - 8 x 8.6 per group)
    'lds_w128': 33.0,     # ds_write_b128 / ds_write2st64 (the store's data path: 135 alone at W = 4, one per ~4 vector instructions here)
    'vmem': 14.0,         # global_load / global_store (address registers ready; not measured separately)
    'mfma': 16.7,         # v_mfma_f32_16x16x32_f16, one wavefront per SIMD issuing them (the T role): the pipe's 16.5 clocks
    'barrier': 0.0,
}
MEASURED = {              # clocks per tick, median of the middle ticks of profiles/r03q_timeline_final.txt
    'T (tabulator, wavefronts 0-2)': (2800 + 630, 'mfma+table 2790-3200, amplitudes / descriptors / barrier bookkeeping 340-650'),
    'T3 (+ phase tables)': (3100 + 1090, 'mfma+table 3100, phase tables 1090'),
    'B (interpolator, 4 tiles)': (3050 + 150 + 105, 'phase B 2440-3360, top 120-200, end 105'),
    'A (row maker, 4 row pairs)': (2750 + 950 + 60, 'phase A 2500-3060, descriptor + next rows issued 770-1130 (starved at the top of the tick: the youngest wavefronts)'),
}


def classify(op):
  if op.startswith('v_mfma'):
    return 'mfma'
  if op.startswith('s_barrier'):
    return 'barrier'
  if op.startswith('s_waitcnt') or op.startswith('s_nop'):
    return 'swait'
  if op.startswith('s_cbranch') or op.startswith('s_branch'):
    return 'branch'
  if op.startswith('s_'):
    return 'salu'
  if op.startswith('ds_write_b128') or op.startswith('ds_write2st64') or op.startswith('ds_write2_b64'):
    return 'lds_w128'
  if op.startswith('ds_'):
    return 'lds'
  if op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'):
    return 'vmem'
  if op.endswith('_dpp'):
    return 'dpp'
  if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
    return 'readlane'
  if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32', op) or op.startswith('v_permlane'):
    return 'trans'
  if (re.match(r'v_(cvt|floor|fract|ceil|rndne|trunc)_', op) or op.startswith('v_pk_') or op.endswith('_f64') or
      op.endswith('_f64_e32') or op.endswith('_f64_e64') or op.startswith('v_lshl_add') or op.startswith('v_mad_') or
      op.startswith('v_mul_lo') or op.startswith('v_mul_hi') or op.startswith('v_med3') or op.startswith('v_bfi') or
      op.startswith('v_fma_mix') or op.startswith('v_add3') or op.startswith('v_and_or') or op.startswith('v_lshl_or') or
      op.startswith('v_alignbit') or op.startswith('v_perm_b32') or op.startswith('v_cmp') and '_f64' in op):
    return 'class2'
  return 'simple'


def blocks_of(body):
  out, cur = [], []
  for line in body.split('\n'):
    s = line.strip()
    if re.match(r'^\.LBB\d+_\d+:', s) or s.startswith('; %bb.'):
      if cur:
        out.append(cur)
      cur = []
      continue
    if not s or s.startswith(';') or s.startswith('.'):
      continue
    op = s.split()[0]
    cur.append(op)
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_barrier'):
      out.append(cur)
      cur = []
  if cur:
    out.append(cur)
  return out


def count(ops, pred):
  return sum(1 for o in ops if pred(o))


def main():
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  from ddsp_amd import build
  flags = [f for f in build.FLAGS if f not in ('-shared', '-fPIC')]
  with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'ht.s')
    subprocess.run([hipcc] + flags + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', out,
                                      os.path.join(ROOT, 'ddsp_amd', 'csrc', 'harmonic_table.hip')], check=True,
                   stderr=subprocess.DEVNULL)
    asm = open(out).read()
  m = re.search(r'^_ZN4ddsp17harm_table_kernelILi6ELi2ELb1ELb0ELb1E\w+:.*?\n(.*?)\n\s*s_endpgm', asm, re.S | re.M)
  blocks = blocks_of(m.group(1))
  nb = lambda b, p: count(b, lambda o: o.startswith(p))
  # the hot blocks of a tick, by what they contain
  b_tiles = max((b for b in blocks if nb(b, 'ds_read2_b32') >= 24), key=len)            # phase B, four tiles
  b_store = next(b for b in blocks if nb(b, 'global_store') == 3 and len(b) < 40)      # its stores (the fourth rides in the block above)
  a_main = [b for b in blocks if nb(b, 'v_exp_f32') >= 32][-1]                          # phase A, four row pairs (the rotated loop body)
  a_write = [b for b in blocks if nb(b, 'ds_write') == 8 and len(b) > 60][-1]           # the splits and plane writes
  a_fetch = [b for b in blocks if nb(b, 'global_load') == 8 and len(b) < 75][-1]          # the top of a tick: the descriptor, the rows of the tick after
  t_mfma = [b for b in blocks if nb(b, 'v_mfma') == 24]                                 # two row tiles
  t_tables = max((b for b in blocks if nb(b, 'v_fma_f64') + nb(b, 'v_add_f64') + nb(b, 'v_mul_f64') >= 6 and nb(b, 'ds_read2_b32') == 0), key=len)
  loop_glue = 45          # per tick and wavefront: descriptor read / take (4 v_readfirstlane), the shifts, the dispatch on the tile count
  roles = {
      'T (tabulator, wavefronts 0-2)': [t_mfma[0], t_mfma[1], ['s_add_u32'] * loop_glue],
      'T3 (+ phase tables)': [t_mfma[0], t_mfma[1], t_tables, ['s_add_u32'] * loop_glue],
      'B (interpolator, 4 tiles)': [b_tiles, b_store, ['s_add_u32'] * loop_glue],
      'A (row maker, 4 row pairs)': [a_fetch, a_main, a_write, ['s_add_u32'] * (loop_glue - 20)],
  }
  classes = list(PRICE)
  print('harm_table_kernel<6, 2, true, false, true> (K = 100, hop 64): instructions per tick and wavefront, by class; price = clocks as ONE wavefront')
  print('sees them at four wavefronts per SIMD (profiles/r03a_*, r03b_*); predicted = sum; measured = its part of the tick (profiles/r03q_*)')
  print()
  print('%-32s' % 'role' + ''.join('%9s' % c for c in classes if c != 'barrier') + '   total  predicted  measured   ratio')
  print('%-32s' % 'price (clocks / instruction)' + ''.join('%9.1f' % PRICE[c] for c in classes if c != 'barrier'))
  simd_time = 0.0
  for role, bl in roles.items():
    ops = [o for b in bl for o in b]
    n = {c: count(ops, lambda o, c=c: classify(o) == c) for c in classes}
    pred = sum(n[c] * PRICE[c] for c in classes)
    meas, note = MEASURED[role]
    print('%-32s' % role + ''.join('%9d' % n[c] for c in classes if c != 'barrier') +
          '  %6d  %9.0f  %8d   %5.2f   (%s)' % (len(ops), pred, meas, pred / meas, note))
    weight = {'T (tabulator, wavefronts 0-2)': 3, 'T3 (+ phase tables)': 1, 'B (interpolator, 4 tiles)': 8,
              'A (row maker, 4 row pairs)': 4}[role]
    simd_time += weight * pred / 4.0             # a wavefront-instruction's SIMD time at W = 4 is a quarter of its wave price
  print()
  print('SIMD view: sum over the 16 wavefronts of predicted / 4 = %.0f clocks of issue time per tick on the 4 SIMDs = %.0f per SIMD;'
        % (simd_time, simd_time / 4))
  print('the tick is ~4300 clocks (37.7 us / 19 ticks, batch 128; 4950 with the stamps of the timeline build): the SIMDs issue %.0f %% of'
        % (100 * simd_time / 4 / 4300))
  print('the time.  A wavefront is not bound by its own instruction count any more (r03o: 20 % fewer row-maker instructions, 9 % fewer')
  print('interpolator instructions: 1.7 % of the kernel): the SIMD serves its four wavefronts oldest first, the row makers (the')
  print('youngest) get what is left and arrive last; what a role costs the kernel is its ablation (profiles/r03o_ablation_roles.txt:')
  print('tabulators 7.6 us, interpolators 8.9, row makers 4.2 of 37.9), not its share of the instructions.')


if __name__ == '__main__':
  main()
