"""Multi-GPU checks that need >= 2 GPUs on the node (they skip on the 1-GPU boxes the round's tests run on): the N > 1
path of bench.py over RCCL (`--gpus 2 --allgather`), and batch shards + all_gather of the audio against the rows one
GPU makes from the whole batch (ddsp_amd/distributed.py; the reference scales the same way, per-replica batch shards:
ddsp/training/trainers.py:145-160).  The same logic runs on CPU with gloo in tests/test_distributed_cpu.py and
tests/test_bench_contract.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two_gpus():
  if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs on the node (found %d)' % (torch.cuda.device_count() if torch.cuda.is_available() else 0))


def _env():
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  return env


def test_bench_two_gpus_allgather():
  _need_two_gpus()
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--allgather', '--steps', '5',
                      '--warmup', '2', '--batch', '8', '--no-cpu-baseline', '--no-aux', '--no-second-shape'],
                     env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1
  line = json.loads(lines[0])
  assert line['n_gpus'] == 2 and line['config']['global_batch'] == 16 and line['scaling'] == 'weak'
  assert line['value'] > 0 and line['allgather_ms'] > 0


def test_two_gpu_shards_gather_to_the_single_gpu_rows():
  _need_two_gpus()
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                      '--master-addr', '127.0.0.1', '--master-port', str(port),
                      os.path.join(ROOT, 'tests', '_two_gpu_gather_rank.py')],
                     env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
  assert r.returncode == 0 and 'TWO_GPU_GATHER OK' in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
