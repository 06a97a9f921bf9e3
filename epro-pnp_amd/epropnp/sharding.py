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


class ObjectExchange:
    """The Det step's whole exchange as ONE collective that stays off the step's critical path.

    `start(local, scalars)` packs this rank's per-object outputs (object axis first, equally padded chunk) and a few
    per-rank scalars (the detection loss's `norm_factor` input, `monte_carlo_pose_loss.py:53` of EPro-PnP-Det) into one
    send buffer and issues ONE `all_gather_into_tensor` on a side stream: the caller's stream goes on with the loss
    and the backward.  `world_mean()` (device tensor, mean over ranks of the scalars) and `objects()` (the full
    `(num_obj, ...)` tensor) make the CURRENT stream wait for the side stream -- an event wait on the device, the host
    never blocks.  Buffers are allocated once per shape and reused (nothing is allocated per step, so the exchange can
    sit inside a hipGraph capture of the step).  A `MonteCarloPoseLoss` accepts the exchange in place of its
    `norm_factor` argument and takes `world_mean()` instead of issuing its own all-reduce.
    Without a process group (or with one rank and `force_collective=False`) no collective is issued."""

    def __init__(self, num_obj, group=None, force_collective=False):
        self.num_obj, self.group, self.force = int(num_obj), group, bool(force_collective)
        self._key = None
        self._side = None
        self._done = None
        self._local = self._scal = None
        self.disabled = False       # timing A/B only (bench.py): the step without its exchange

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _active(self):
        return dist.is_initialized() and (self._world() > 1 or self.force) and not self.disabled

    def start(self, local, scalars=None):
        """local (n_local, ...) per-object outputs of this rank's shard; scalars: tensor / float / sequence of them."""
        world = self._world()
        local = local.detach()
        scal = None
        if scalars is not None:
            scal = torch.as_tensor(scalars, dtype=local.dtype, device=local.device).detach().reshape(-1)
        self._local, self._scal = local, scal
        self._n_scal = 0 if scal is None else scal.numel()
        if not self._active():
            return self
        chunk = (self.num_obj + world - 1) // world
        row = local[0].numel() if local.shape[0] else int(torch.Size(local.shape[1:]).numel())
        key = (world, chunk, tuple(local.shape[1:]), self._n_scal, local.dtype, local.device)
        if key != self._key:
            self._key = key
            self._send = local.new_zeros(chunk * row + self._n_scal)
            self._recv = local.new_empty(world * (chunk * row + self._n_scal))
            if local.is_cuda:
                self._side = torch.cuda.Stream(device=local.device)
                self._done = torch.cuda.Event()
        self._chunk, self._row = chunk, row
        n = local.shape[0] * row
        if local.is_cuda:
            cur = torch.cuda.current_stream(local.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._send[:n].copy_(local.reshape(-1))
                if self._n_scal:
                    self._send[chunk * row:].copy_(scal)
                dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
                self._done.record(self._side)
            for t in (local, scal):         # consumed on the side stream: keep the allocator from recycling them early
                if t is not None and not torch.cuda.is_current_stream_capturing():
                    t.record_stream(self._side)
        else:
            self._send[:n].copy_(local.reshape(-1))
            if self._n_scal:
                self._send[chunk * row:].copy_(scal)
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        return self

    def _wait(self):
        if self._done is not None and self._local.is_cuda:
            torch.cuda.current_stream(self._local.device).wait_event(self._done)

    def world_mean(self):
        """Mean over ranks of the scalars handed to start(): shape (n_scalars,), or () for a single scalar."""
        assert self._scal is not None, 'start() was called without scalars'
        if not self._active():
            return self._scal.reshape(()) if self._n_scal == 1 else self._scal
        self._wait()
        world = self._world()         # mmdet's reduce_mean arithmetic: divide by the world size, then sum over ranks
        m = self._recv.view(world, -1)[:, self._chunk * self._row:].div(world).sum(0)
        return m.reshape(()) if self._n_scal == 1 else m

    def objects(self):
        """The gathered (num_obj, ...) per-object outputs, ranks in order, padding trimmed."""
        if not self._active():
            return self._local
        self._wait()
        world = self._world()
        per = self._recv.view(world, -1)[:, :self._chunk * self._row]
        pieces = []
        for r in range(world):
            lo, hi = shard_range(self.num_obj, r, world)
            pieces.append(per[r, :(hi - lo) * self._row])
        return torch.cat(pieces, 0).view((self.num_obj,) + tuple(self._local.shape[1:]))
