"""Multi-GPU layout of the hot path: one process per GPU, the batch sharded contiguously.

Batch rows never interact on this path (every reduction / scan is over harmonics or time,
SURVEY.md 8e; the reference scales the same way: tf.distribute per-replica batch shards,
ddsp/training/trainers.py:145-160), so there is NO data-path collective.  The only optional
communication is a final all-gather of the synthesised audio over RCCL / xGMI.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world_size):
  """Contiguous [start, stop) rows of `rank`; the first `global_batch % world_size` ranks get one more."""
  if not 0 <= rank < world_size:
    raise ValueError('rank {} outside world of {}'.format(rank, world_size))
  base, extra = divmod(int(global_batch), int(world_size))
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank=None, world_size=None):
  """Slice every [B, ...] tensor of a dict / tuple / tensor down to this rank's rows."""
  if rank is None:
    rank = dist.get_rank() if dist.is_initialized() else 0
  if world_size is None:
    world_size = dist.get_world_size() if dist.is_initialized() else 1

  def one(t):
    lo, hi = shard_bounds(t.shape[0], rank, world_size)
    return t[lo:hi]
  if isinstance(tensors, dict):
    return {k: one(v) for k, v in tensors.items()}
  if isinstance(tensors, (tuple, list)):
    return type(tensors)(one(v) for v in tensors)
  return one(tensors)


def all_gather_audio(audio, global_batch=None):
  """Optional epilogue: every rank receives the whole [global_batch, n_samples] audio.

  Uses all_gather_into_tensor when shards are equal (one RCCL call; each 4*B/G*N-byte shard
  crosses one xGMI link), falling back to all_gather with padding for ragged shards.
  """
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return audio
  world = dist.get_world_size()
  local = audio.shape[0]
  if global_batch is None:
    counts = torch.tensor([local], dtype=torch.int64, device=audio.device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    sizes = [int(c.item()) for c in all_counts]
  else:
    sizes = [shard_bounds(global_batch, r, world)[1] - shard_bounds(global_batch, r, world)[0]
             for r in range(world)]
  audio = audio.contiguous()
  if len(set(sizes)) == 1:
    out = torch.empty((sum(sizes),) + tuple(audio.shape[1:]), dtype=audio.dtype,
                      device=audio.device)
    dist.all_gather_into_tensor(out, audio)
    return out
  biggest = max(sizes)
  padded = torch.zeros((biggest,) + tuple(audio.shape[1:]), dtype=audio.dtype,
                       device=audio.device)
  padded[:local] = audio
  parts = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(parts, padded)
  return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
