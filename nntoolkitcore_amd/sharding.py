"""Utterance-level sharding across the GPUs of one node.

The inference path has no cross-utterance dependency (BatchNorm uses stored moving
statistics, every sequence owns its recurrent state), so the batch dimension is split
into contiguous shards, one process per GPU, with NO collective on the data path.  The
only communication is one broadcast of the packed weight blob from rank 0 at start-up
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
import numpy as np


def shard_range(n_utterances, world, rank):
    """Contiguous, balanced [lo, hi) of utterances owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(n_utterances, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(flat, torch, dist, src=0):
    """Broadcast one flat fp32 weight vector from `src` to every rank; returns a numpy array.
    With the nccl (= RCCL) backend the buffer lives in HBM and travels over xGMI."""
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    on_gpu = dist.get_backend() == "nccl"
    t = torch.from_numpy(flat.copy())
    if on_gpu:
        t = t.cuda()
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_shards(local, torch, dist):
    """Concatenate per-rank output shards along the utterance axis on every rank (test/debug helper;
    a production caller keeps each shard on its GPU)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    return np.concatenate(parts, axis=0)
