"""Which kernels' machine code changed?  Compiles every csrc/*.hip to gfx950 assembly (hipcc -S, device only), hashes
each kernel's instruction stream (symbols, labels and comments normalised away) and compares with a recorded baseline.

    python tools/isa_guard.py --record profiles/r01_isa_hashes.json      # write the baseline
    python tools/isa_guard.py profiles/r01_isa_hashes.json               # list kernels that differ / are new

Used when an opt-in variant (a new template parameter) is added to a measured kernel: the default instantiation
must come out instruction for instruction as before, whatever its mangled name now is."""
import hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ddsp_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-I' + os.path.join(ROOT, 'include'),
         '--cuda-device-only', '-S']


def demangled_kernels(asm_path):
  txt = open(asm_path).read()
  out = {}
  for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)\n\.Lfunc_end\d+:', txt, re.S | re.M):
    body = re.sub(r'_Z\w+', 'SYM', m.group(2))
    body = re.sub(r'\.L\w+', 'LBL', body)
    body = '\n'.join(line.split(';')[0].rstrip() for line in body.splitlines())
    out[m.group(1)] = hashlib.sha1(body.encode()).hexdigest()[:16]
  if not out:
    return out
  names = subprocess.run(['c++filt'] + list(out), capture_output=True, text=True).stdout.split('\n')
  return {re.sub(r'\(.*', '', n): h for n, h in zip(names, out.values())}


def current():
  hashes = {}
  with tempfile.TemporaryDirectory() as tmp:
    for name in sorted(os.listdir(CSRC)):
      if not name.endswith('.hip'):
        continue
      asm = os.path.join(tmp, name + '.s')
      subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [os.path.join(CSRC, name), '-o', asm], check=True,
                     stderr=subprocess.DEVNULL)
      for kernel, h in demangled_kernels(asm).items():
        hashes['%s: %s' % (name, kernel)] = h
  return hashes


if __name__ == '__main__':
  if len(sys.argv) == 3 and sys.argv[1] == '--record':
    json.dump(current(), open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    print('recorded', sys.argv[2])
  else:
    base = json.load(open(sys.argv[1]))
    now = current()
    # a default instantiation keeps its hash when a defaulted template parameter is appended to its name
    base_hashes = set(base.values())
    changed = [k for k, h in now.items() if k in base and base[k] != h]
    new = [k for k, h in now.items() if k not in base and h not in base_hashes]
    renamed = [k for k, h in now.items() if k not in base and h in base_hashes]
    gone = [k for k, h in base.items() if k not in now and h not in set(now.values())]
    print('%d kernels, %d unchanged, %d renamed with identical code' % (len(now), len(now) - len(changed) - len(new) - len(renamed), len(renamed)))
    for title, items in (('CHANGED', changed), ('new', new), ('gone', gone)):
      for k in items:
        print('  %-8s %s' % (title, k))
    sys.exit(1 if changed or gone else 0)
