"""TEST INFRASTRUCTURE ONLY: builds ddsp_amd/csrc/general.hip for the HOST with g++ against the stand-in
<hip/hip_runtime.h> of tests/hip_emu/include (a kernel launch becomes a serial loop over blocks and threads)
and loads the result with the argument types of ddsp_amd/_lib.py.  The CPU test run uses it to check the index
arithmetic and launch geometry of those plain-HIP kernels against the oracle; nothing in ddsp_amd/ ever loads
it, and the `-m gpu` tests run the same checks on the real library through the same C ABI."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

from ddsp_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCE = os.path.join(ROOT, 'ddsp_amd', 'csrc', 'general.hip')
OUT = os.path.join(HERE, '_build', 'libddsp_general_emu.so')
ENTRY_POINTS = ['ddsp_resample_ex_f32', 'ddsp_fft_convolve_f32', 'ddsp_harmonic_envelopes_f32',
                'ddsp_scale_f32', 'ddsp_harmonic_oscillator_bank_workspace_bytes', 'ddsp_harmonic_oscillator_bank_f32',
                'ddsp_harmonic_f0_grad_workspace_bytes', 'ddsp_harmonic_f0_grad_f32', 'ddsp_exp_decay_ir_f32',
                'ddsp_exp_decay_ir_backward_workspace_bytes', 'ddsp_exp_decay_ir_backward_f32', 'ddsp_sigmoid_f32',
                'ddsp_mix_f32', 'ddsp_sigmoid_backward_f32', 'ddsp_mix_backward_f32', 'ddsp_resample_ex_backward_f32']

_emu = None


def _digest():
  h = hashlib.sha256()
  for path in (SOURCE, os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'),
               os.path.join(ROOT, 'include', 'ddsp_amd.h')):
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def load():
  """The host build of general.hip, entry points typed as in ddsp_amd/_lib.py."""
  global _emu
  if _emu is not None:
    return _emu
  stamp = OUT + '.stamp'
  digest = _digest()
  os.makedirs(os.path.dirname(OUT), exist_ok=True)
  import fcntl
  with open(OUT + '.lock', 'w') as lock:                # (pytest-xdist workers that find the library stale at the same moment)
    fcntl.flock(lock, fcntl.LOCK_EX)
    if not (os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == digest):
      tmp = OUT + '.tmp.%d' % os.getpid()
      subprocess.run(['g++', '-x', 'c++', '-std=c++17', '-O2', '-ffp-contract=off', '-shared', '-fPIC',
                      '-I' + os.path.join(HERE, 'include'), '-I' + os.path.join(ROOT, 'include'), SOURCE,
                      '-o', tmp], check=True)
      os.replace(tmp, OUT)
      with open(stamp, 'w') as f:
        f.write(digest + '\n')
  lib = ctypes.CDLL(OUT)
  for name in ENTRY_POINTS:
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = _lib.SIGNATURES[name]
  _emu = lib
  return lib


def f32(x):
  return np.ascontiguousarray(x, dtype=np.float32)


def ptr(a):
  """Host pointer of a numpy array (or None), in the slot where the product passes a device pointer."""
  return None if a is None else a.ctypes.data
