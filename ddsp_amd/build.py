"""Build libddsp_amd.so for gfx950 with hipcc (in-tree, so it ships to the GPU box)."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ['harmonic.hip', 'harmonic_table.hip', 'harmonic_bwd_table.hip', 'filtered_noise.hip', 'filtered_noise_mfma.hip', 'filtered_noise_general.hip', 'reverb.hip', 'spectral_loss.hip', 'spectral_terms.hip', 'general.hip', 'profile.hip']
OUT = os.path.join(HERE, 'lib', 'libddsp_amd.so')


STAMP = OUT + '.stamp'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-fno-slp-vectorize']
# Per-source compiler switches.  The FIR kernels of FilteredNoise sit AT their register budget (128 VGPRs, two blocks of eight
# wavefronts per CU); with LLVM's AMDGPU-specific register-pressure trackers the pre-RA scheduler orders their software pipelines
# so that the 2^23-level instance of noise_mfma65_kernel runs 36.6 -> 34.5 us per launch at batch 128 and tv_fir_mfma_kernel
# 35.0 -> 33.5 (same-session A/B, profiles/r06_compiler_scheduling_switches.txt).  Every other source measured the same with it
# or 0.3 us slower (harmonic_table.hip), so it stays per file.
EXTRA_FLAGS = {
    'filtered_noise_mfma.hip': ['-mllvm', '-amdgpu-use-amdgpu-trackers=1'],
    'filtered_noise_general.hip': ['-mllvm', '-amdgpu-use-amdgpu-trackers=1'],
}


def _deps():
  deps = sorted(os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc')))
  deps.append(os.path.join(ROOT, 'include', 'ddsp_amd.h'))
  return deps


def source_digest():
  """sha256 over the sources, the header and the compiler flags: a snapshot copied to another
  box need not preserve mtimes, and a needless rebuild there costs minutes of GPU-box time."""
  h = hashlib.sha256(' '.join(FLAGS + SOURCES + [k + ' '.join(v) for k, v in sorted(EXTRA_FLAGS.items())]).encode())
  for d in _deps():
    with open(d, 'rb') as f:
      h.update(os.path.basename(d).encode() + b'\0' + f.read())
  return h.hexdigest()


def needs_rebuild():
  if not os.path.exists(OUT):
    return True
  if os.path.exists(STAMP):
    with open(STAMP) as f:
      return f.read().strip() != source_digest()
  t = os.path.getmtime(OUT)
  return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True):
  if not force and not needs_rebuild():
    return OUT
  hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  os.makedirs(os.path.dirname(OUT), exist_ok=True)
  objdir = os.path.join(HERE, 'lib', 'obj')
  os.makedirs(objdir, exist_ok=True)
  compile_flags = [f for f in FLAGS if f != '-shared'] + ['-I' + os.path.join(ROOT, 'include')]

  def compile_one(src):
    obj = os.path.join(objdir, src.replace('.hip', '.o'))
    cmd = [hipcc] + compile_flags + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(HERE, 'csrc', src), '-o', obj]
    if verbose:
      print('[ddsp_amd.build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj

  # one hipcc per translation unit, side by side (the sources are independent: no relocatable device code)
  from concurrent.futures import ThreadPoolExecutor
  with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
    objs = list(pool.map(compile_one, SOURCES))
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT]
  if verbose:
    print('[ddsp_amd.build]', ' '.join(cmd), flush=True)
  subprocess.run(cmd, check=True)
  with open(STAMP, 'w') as f:
    f.write(source_digest() + '\n')
  return OUT


if __name__ == '__main__':
  build(force='--force' in sys.argv)
