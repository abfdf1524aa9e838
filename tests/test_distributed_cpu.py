"""world_size-2 gloo test (CPU) of the multi-GPU layout: contiguous batch shards, no data-path
collective, optional all-gather of the audio.  The per-rank "synthesis" here is the CPU oracle
(this is a tests/ file); on GPUs each rank runs the HIP path on exactly these shards."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, global_batch, ragged, out_dir):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from ddsp_amd import distributed as D
  from oracle import ddsp_oracle as O
  rng = np.random.default_rng(0)                    # every rank builds the same global inputs
  F, K, N = 6, 5, 384
  x = dict(amplitudes=torch.from_numpy(rng.standard_normal((global_batch, F, 1)).astype(np.float32)),
           harmonic_distribution=torch.from_numpy(rng.standard_normal((global_batch, F, K)).astype(np.float32)),
           f0_hz=torch.from_numpy((200 + rng.standard_normal((global_batch, F, 1))).astype(np.float32)))
  mine = D.shard_batch(x)
  lo, hi = D.shard_bounds(global_batch, rank, world)
  assert mine['f0_hz'].shape[0] == hi - lo
  local = torch.from_numpy(O.harmonic(mine['amplitudes'].numpy(), mine['harmonic_distribution'].numpy(),
                                      mine['f0_hz'].numpy(), N).astype(np.float32))
  full = D.all_gather_audio(local, None if ragged else global_batch)
  ref = O.harmonic(x['amplitudes'].numpy(), x['harmonic_distribution'].numpy(), x['f0_hz'].numpy(), N)
  np.testing.assert_array_equal(full.numpy(), ref.astype(np.float32))      # rows independent: bit exact
  dist.barrier()
  dist.destroy_process_group()
  open(os.path.join(out_dir, 'ok%d' % rank), 'w').write('ok')


@pytest.mark.parametrize('global_batch,ragged', [(4, False), (5, True)])
def test_batch_shards_and_all_gather_world2(tmp_path, global_batch, ragged):
  port = 29500 + (os.getpid() % 2000) + global_batch
  mp.spawn(_worker, args=(2, port, global_batch, ragged, str(tmp_path)), nprocs=2, join=True)
  assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_shard_bounds_cover_the_batch():
  from ddsp_amd import distributed as D
  for world in (1, 2, 3, 8):
    for n in (1, 7, 8, 32, 1024):
      spans = [D.shard_bounds(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    D.shard_bounds(8, 8, 8)
