"""Object-level sharding of one batch over the GPUs of a node (one process per GPU, RCCL over xGMI).

Every object is independent through the LM solve, the AMIS sampler, the per-object loss and the backward
(SURVEY.md section 8e), so a batch shards embarrassingly: contiguous split of the object axis, no collective on the
data path.  The only exchange is an all-gather of per-object OUTPUTS when one batch of detections was split across
ranks (EPro-PnP-Det inference path); it is a single `all_gather_into_tensor` on contiguous, equally padded chunks --
latency-bound (KB..MB), far below the per-link xGMI budget, so no ring/bucketing logic is warranted.
The reference has no counterpart (it runs DDP over images and never splits a batch of objects).
"""
import torch
import torch.distributed as dist


def shard_range(num_obj, rank=None, world_size=None):
    """[start, stop) of this rank's contiguous shard; the first `num_obj % world_size` ranks get one extra object."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, extra = divmod(num_obj, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_objects(tensors, num_obj, obj_dim=0, rank=None, world_size=None):
    """Slice each tensor (or None) along its object axis to this rank's shard."""
    lo, hi = shard_range(num_obj, rank, world_size)
    out = []
    for t in tensors:
        out.append(None if t is None else t.narrow(obj_dim, lo, hi - lo))
    return out


def gather_objects(local, num_obj, obj_dim=0, group=None, force_collective=False):
    """Inverse of shard_objects for an output tensor: every rank receives the full (num_obj, ...) tensor.
    One `all_gather_into_tensor` over equally padded per-rank chunks (uneven tails are padded, then trimmed).
    A single-rank group returns `local` untouched unless `force_collective` asks for the (trivial) collective anyway --
    bench.py does, so that the very code path of the 8-GPU run executes under `torchrun --nproc-per-node 1`."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return local
    world = dist.get_world_size(group)
    chunk = (num_obj + world - 1) // world
    x = local.movedim(obj_dim, 0).contiguous()
    pad = chunk - x.shape[0]
    if pad:
        x = torch.cat((x, x.new_zeros((pad,) + x.shape[1:])), 0)
    out = x.new_empty((world * chunk,) + x.shape[1:])
    dist.all_gather_into_tensor(out, x, group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(num_obj, r, world)
        pieces.append(out[r * chunk: r * chunk + (hi - lo)])
    return torch.cat(pieces, 0).movedim(0, obj_dim)
