"""TEST INFRASTRUCTURE ONLY: builds the gfx950 kernel sources of ddsp_amd/csrc for the HOST with the ROCm clang++
against the SIMT stand-in <hip/hip_runtime.h> of tests/hip_emu/include_simt (threads of a block as fibers,
wavefront operations evaluated when all live lanes have arrived, MFMA / DPP / swizzle modelled) and loads the
result with the argument types of ddsp_amd/_lib.py.  The sources are compiled as they are, with two textual
substitutions the host language needs:
  * `extern __shared__ <attrs> T name[];`   ->  a pointer to the emulated block's dynamic LDS;
  * the four `asm volatile(...)` statements of harmonic.hip (pairs of scalar row loads with their wait, one
    `s_waitcnt vmcnt(0)`) -> the same loads / a wavefront join in C++.
Nothing in ddsp_amd/ ever loads this; it exists so that kernel logic can be checked on the CPU when no GPU
(or no GPU budget) is at hand."""
import ctypes
import hashlib
import os
import re
import subprocess

from ddsp_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'ddsp_amd', 'csrc')
BUILD = os.path.join(HERE, '_build', 'simt')
OUT = os.path.join(BUILD, 'libddsp_simt_emu.so')
SOURCES = ['harmonic.hip', 'harmonic_table.hip', 'harmonic_bwd_table.hip', 'filtered_noise.hip', 'filtered_noise_mfma.hip', 'filtered_noise_general.hip', 'reverb.hip', 'spectral_loss.hip', 'spectral_terms.hip', 'general.hip',
           'profile.hip']
CLANG = '/opt/rocm/lib/llvm/bin/clang++'

_DYN_LDS = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w:]*)\s+(\w+)\[\];')

_emu = None


# the scalar loads of the direct-sum kernel: two rows of N dwords into SGPR vectors, then their wait
_SLOAD = re.compile(
    r'asm volatile\("s_load_dwordx(\d+) %0, %2, (?:0x0|%4)\\n\\ts_load_dwordx\d+ %1, %3, (?:0x0|%4)\\n\\ts_waitcnt lgkmcnt\(0\)"'
    r'\s*:\s*"=&s"\((\w+)\),\s*"=&s"\((\w+)\)\s*:\s*"s"\((\w+)\),\s*"s"\((\w+)\)(?:,\s*"i"\((\w+)\))?\s*:\s*"memory"\);')
_WAITCNT = re.compile(r'asm volatile\("s_waitcnt vmcnt\(0\)"\s*:::\s*"memory"\);')


def _preprocess(text):
  text = text.replace('"../../include/ddsp_amd.h"', '<ddsp_amd.h>')      # the staged copies live elsewhere
  text = _SLOAD.sub(lambda m: 'ddsp_emu_sload<%s>(%s, %s, %s, %s, %s);' % (
      m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6) or '0'), text)
  text = _WAITCNT.sub('ddsp_emu_wave_sync();', text)
  if 'asm volatile' in text:
    raise RuntimeError('inline assembly the emulation does not know')
  return _DYN_LDS.sub(r'DDSP_EMU_DYNAMIC_LDS(\1, \2);', text)


def _digest():
  h = hashlib.sha256(' '.join(SOURCES).encode())
  paths = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
  paths += [os.path.join(HERE, 'include_simt', 'hip', 'hip_runtime.h'), os.path.join(ROOT, 'include', 'ddsp_amd.h'),
            os.path.abspath(__file__)]
  for path in paths:
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def build(verbose=False):
  """Builds under an exclusive lock on the build directory: pytest-xdist workers that find the library stale at the same
  moment used to stage and compile into the same files at once (one of them then failed to load a half-written object)."""
  import fcntl
  os.makedirs(BUILD, exist_ok=True)
  with open(os.path.join(BUILD, '.lock'), 'w') as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    return _build_locked(verbose)


def _build_locked(verbose):
  stamp = OUT + '.stamp'
  digest = _digest()
  if os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
    return OUT
  staged = []
  for name in os.listdir(CSRC):                       # headers travel with the sources
    with open(os.path.join(CSRC, name)) as f:
      text = f.read()
    dst = os.path.join(BUILD, name.replace('.hip', '.cpp'))
    with open(dst, 'w') as f:
      f.write(_preprocess(text))
    if name in SOURCES:
      staged.append(dst)
  # -ffp-contract=fast -mfma: a*b+c contracts to an FMA as in hipcc's default mode (the FFT butterflies of the
  # SpectralLoss are measurably less accurate without); fast-honor-pragmas is hipcc's own default, so the rn_* helpers of common.h stay uncontracted here exactly as on the device
  cmd = [CLANG, '-std=c++17', '-O1', '-g0', '-ffp-contract=fast-honor-pragmas', '-mfma', '-shared', '-fPIC', '-w',
         '-I' + os.path.join(HERE, 'include_simt'), '-I' + os.path.join(ROOT, 'include'),
         '-include', os.path.join(HERE, 'include_simt', 'hip', 'hip_runtime.h')] + staged + ['-o', OUT]
  if verbose:
    print(' '.join(cmd))
  subprocess.run(cmd, check=True)
  with open(stamp, 'w') as f:
    f.write(digest + '\n')
  return OUT


def load():
  global _emu
  if _emu is None:
    lib = ctypes.CDLL(build())
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
      if hasattr(lib, name):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    _emu = lib
  return _emu
