"""One rank of tests/test_gpu_multi.py: Harmonic + FilteredNoise on this rank's batch shard through the HIP library,
all_gather of the audio over RCCL, rank 0 compares every row with the rows a single GPU makes from the whole batch.
Launched by `python -m torch.distributed.run --nproc-per-node 2 ...` (test infrastructure: never imported)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
  torch.cuda.set_device(local)
  dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
  import ddsp_amd as ddsp
  from ddsp_amd import distributed as D
  global_batch, F, K, N = 6, 100, 100, 6400
  rng = np.random.default_rng(7)                    # every rank builds the same global inputs
  x = dict(amplitudes=rng.standard_normal((global_batch, F, 1)).astype(np.float32),
           harmonic_distribution=rng.standard_normal((global_batch, F, K)).astype(np.float32),
           f0_hz=(70 + rng.standard_normal((global_batch, F, 1))).astype(np.float32),
           magnitudes=rng.standard_normal((global_batch, F, 65)).astype(np.float32))
  noise = rng.uniform(-1, 1, (global_batch, N)).astype(np.float32)
  dev = {k: ddsp.core.tf_float32(v) for k, v in x.items()}
  mine = D.shard_batch(dev)
  lo, hi = D.shard_bounds(global_batch, rank, world)
  harm, fn = ddsp.synths.Harmonic(n_samples=N), ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
  h = harm(mine['amplitudes'], mine['harmonic_distribution'], mine['f0_hz'])
  z = fn(mine['magnitudes'], noise=ddsp.core.tf_float32(noise[lo:hi]))
  full_h = D.all_gather_audio(h, global_batch)
  full_z = D.all_gather_audio(z, global_batch)
  ok = True
  if rank == 0:
    ref_h = harm(dev['amplitudes'], dev['harmonic_distribution'], dev['f0_hz'])
    ref_z = fn(dev['magnitudes'], noise=ddsp.core.tf_float32(noise))
    ok = bool(torch.equal(full_h, ref_h) and torch.equal(full_z, ref_z))      # rows are independent: bit exact
    print('TWO_GPU_GATHER %s' % ('OK' if ok else 'MISMATCH'), flush=True)
  dist.barrier()
  dist.destroy_process_group()
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
