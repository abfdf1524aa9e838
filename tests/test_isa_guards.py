"""Build-time guards on the gfx950 instruction stream of the default Harmonic kernels (common.h:166).

`load_issue` / `loads_landed` (csrc/common.h) hide an in-flight global load from the compiler: the inline-asm output is
"defined" at the issue, the data arrives later, and only the `s_waitcnt vmcnt(0)` of `loads_landed` makes it safe to
touch.  Nothing in the language stops the compiler from copying or spilling such a register in between (under higher
register pressure, or with another hipcc) - it would silently capture stale data.  This test compiles
csrc/harmonic_table.hip to assembly and checks, for every pinned load, that no instruction between the load and the
wait that covers it names the destination registers; and that the kernels use no scratch memory.  CPU only (hipcc
cross-compiles), a few seconds."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _regs(text):
  out = set()
  for lo, hi in re.findall(r'\bv\[(\d+):(\d+)\]', text):
    out.update(range(int(lo), int(hi) + 1))
  out.update(int(r) for r in re.findall(r'\bv(\d+)\b', text))
  return out


@pytest.fixture(scope='module')
def asm(tmp_path_factory):
  if not os.path.exists(HIPCC):
    pytest.skip('hipcc not found')
  out = tmp_path_factory.mktemp('isa') / 'harmonic_table.s'
  from ddsp_amd import build
  flags = [f for f in build.FLAGS if f not in ('-shared', '-fPIC')]
  subprocess.run([HIPCC] + flags + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', str(out),
                                    os.path.join(ROOT, 'ddsp_amd', 'csrc', 'harmonic_table.hip')],
                 check=True, stderr=subprocess.DEVNULL)
  return out.read_text()


def _kernels(asm_text):
  """{mangled name: body text} of the harm_wt16 / harm_table kernels."""
  found = {}
  for m in re.finditer(r'^(_ZN4ddsp\d+harm_(?:wt16|table)_kernel\w+):.*?\n(.*?)\n\s*s_endpgm', asm_text, re.S | re.M):
    found[m.group(1)] = m.group(2)
  return found


def test_no_scratch_in_the_harmonic_table_kernels(asm):
  names = re.findall(r'\.name:\s+(_ZN4ddsp\d+harm_(?:wt16|table)_kernel\w+)', asm)
  assert len(names) >= 6
  for name in names:
    block = asm[asm.index('.name:           ' + name) - 400:asm.index('.name:           ' + name) + 400]
    m = re.search(r'\.private_segment_fixed_size:\s+(\d+)', block)
    # the 129 .. 200-harmonic instances with processors.Add fused in (<10, 4, .., ADD = true, ..>) hold the other signal's four
    # samples per lane across phase B on top of ten taps' worth of registers: three of them go to scratch since phase B runs on
    # plain FMAs (round 4) - 12 bytes per lane and tick, nothing next to a tick of 8 us; every other instance: none
    limit = 24 if re.search(r'harm_table_kernelILi10ELi4ELb[01]ELb1E', name) else 0
    assert m and int(m.group(1)) <= limit, '%s uses %s bytes of scratch per lane' % (name, m and m.group(1))


def test_pinned_loads_are_not_touched_before_their_wait(asm):
  """Loads written as volatile assembly (common.h load_issue) are invisible to the compiler's wait counting: nothing may
  name their destination registers before a wait that covers them.  Vector memory operations return in order, so
  `s_waitcnt vmcnt(N)` - the compiler's or the kernel's own - leaves the N youngest of ALL of them in flight (the WIDE
  tabulators take their sixteen fragments four at a time that way)."""
  kernels = _kernels(asm)
  assert kernels, 'kernels not found in the assembly'
  checked = 0
  for name, body in kernels.items():
    lines = body.split('\n')
    in_asm, inflight = False, []           # inflight: [(line number, destination registers of a PINNED load, or an empty set)]
    at_label = {}                          # what is in flight where a forward branch lands (the scan follows the text, not the
                                           # control flow: code behind an unconditional branch is not reached from above it)
    for n, line in enumerate(lines):
      code = line.split(';')[0].strip() if not line.strip().startswith(';;#') else line.strip()
      if code.startswith(';;#ASMSTART'):
        in_asm = True
        continue
      if code.startswith(';;#ASMEND'):
        in_asm = False
        continue
      if code.endswith(':') and not code.startswith(';'):
        inflight = sorted(set(inflight) | set(at_label.pop(code[:-1], [])), key=lambda e: e[0])
        continue
      if not code or code.startswith('.'):
        continue
      if code.startswith('s_cbranch') or code.startswith('s_branch'):
        target = code.split()[-1]
        at_label[target] = sorted(set(at_label.get(target, [])) | set(inflight), key=lambda e: e[0])
        if code.startswith('s_branch'):
          inflight = []
        continue
      if code.startswith('s_waitcnt'):
        m = re.search(r'vmcnt\((\d+)\)', code)
        if m:
          keep = int(m.group(1))
          inflight = inflight[len(inflight) - keep:] if keep else []
        continue
      if code.startswith('s_barrier'):
        continue
      pending = {r: ln for ln, regs in inflight for r in regs}
      is_vmem = code.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))
      if in_asm and code.startswith('global_load'):
        dst = code.split(',')[0]
        touched = _regs(code.split(',', 1)[1]) & set(pending)
        assert not touched, '%s: line %d `%s` uses v%s, still in flight' % (name, n, code, sorted(touched))
        inflight.append((n, frozenset(_regs(dst))))
        checked += 1
        continue
      # any other instruction: must not name a pending destination register (address operands of later pinned loads
      # are covered by the same rule: they are ordinary instructions' results, never the pending registers)
      touched = _regs(code) & set(pending)
      assert not touched, '%s: line %d `%s` touches v%s, the destination of a pinned load still in flight (issued at line %d)' % (
          name, n, code, sorted(touched), min(pending[r] for r in touched))
      if is_vmem:
        inflight.append((n, frozenset()))
  assert checked >= 6          # the pinned loads of the T-wavefront that builds the phase tables, in every instantiation


def test_no_vector_written_scalar_base_right_before_a_pinned_load(asm):
  """A vector instruction that writes a scalar register (v_readlane - how the compiler reads back a scalar it spilled
  to a register lane -, v_readfirstlane, a compare with a scalar destination) must be five wait states ahead of a vector
  memory instruction that reads that register.  The compiler pads its own loads; it does not look inside assembly
  statements, so the pinned loads pad themselves where it matters (common.h load_issue_spaced).  Found on the MI355X
  (profiles/r03u_*): the 129 .. 200-harmonic instances read fragments through stale bases."""
  kernels = _kernels(asm)
  assert kernels
  loads = 0
  for name, body in kernels.items():
    ins, in_asm = [], False
    for line in body.split('\n'):
      t = line.strip()
      if t.startswith(';;#ASMSTART'):
        in_asm = True
        continue
      if t.startswith(';;#ASMEND'):
        in_asm = False
        continue
      code = line.split(';')[0].strip()
      if not code or code.startswith('.') or code.endswith(':'):
        continue
      ins.append((code, in_asm))
    for i, (code, pinned) in enumerate(ins):
      if not (pinned and code.startswith('global_load')):
        continue
      m = re.search(r's\[(\d+):(\d+)\]', code)
      if not m:
        continue                         # (a 64-bit vector address: no scalar operand)
      loads += 1
      base = set(range(int(m.group(1)), int(m.group(2)) + 1))
      states = 0
      for prev, _ in reversed(ins[max(0, i - 8):i]):
        if states >= 5:
          break
        w = re.match(r'v_(?:readlane_b32|readfirstlane_b32|cmp\w*_e64|cndmask\w*|add_co\w*|addc_co\w*)\s+s\[?(\d+)(?::(\d+))?\]?', prev)
        if w:
          written = set(range(int(w.group(1)), int(w.group(2) or w.group(1)) + 1))
          assert not (written & base), '%s: `%s` reads s%s %d wait state(s) after `%s` wrote it' % (name, code, sorted(base), states, prev)
        states += int(prev.split()[1]) + 1 if prev.startswith('s_nop') else 1
  assert loads >= 24



def test_no_assembly_statement_packed_fma_in_phase_b(asm):
  """Round 4 (csrc/harmonic_table.hip, "phase B on pairs of fp32 values"; profiles/r04_packed_fma_glitch.txt): runs of v_pk_fma_f32
  issued from assembly statements in the interpolators came back wrong in the last sixteen lanes now and then while a tabulator of
  the same SIMD ran MFMAs.  Phase B is plain v_fma_f32 since; the statements survive behind -DDDSP_EXP_PACKED_PHASE_B for
  experiments and must not reach the product build."""
  kernels = _kernels(asm)
  assert kernels
  for name, body in kernels.items():
    in_asm = False
    for line in body.split('\n'):
      t = line.strip()
      if t.startswith(';;#ASMSTART'):
        in_asm = True
      elif t.startswith(';;#ASMEND'):
        in_asm = False
      elif in_asm:
        assert not t.startswith('v_pk_'), '%s: `%s` inside an assembly statement' % (name, t)
