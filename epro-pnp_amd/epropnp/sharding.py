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


class RcclComm:
    """An RCCL communicator of our own over the ranks of a torch.distributed group, driven through the library's C API
    (the librccl.so that torch itself has loaded): `all_gather` enqueues ONE ncclAllGather on torch's CURRENT stream.

    torch.distributed's nccl backend runs every collective on an internal stream and chains current -> internal -> current
    with events; on this platform each cross-queue hop costs tens of microseconds of GPU time and the Python / c10d
    wrapper ~50 us of host time -- more than the whole Det step's kernels (tools/exchange_probe.py).  Called directly, the
    collective is one kernel in stream order between the step's own kernels: no hop, ~10 us of host time, and it is
    recorded by a hipGraph capture of the step like any other launch.
    The unique id is created on group rank 0 and broadcast through the torch group (which also proves the group works);
    ncclCommInitRank is collective over the group.
    `lib_path`: the shared library to bind instead of the librccl.so next to torch -- any library with RCCL's five entry points
    (tests/stubs/rccl_stub.c drives this very class with 2 and 8 ranks on CPU tensors, where no multi-GPU node is at hand);
    `device`: where the agreement tensors live (default: the current HIP device, or the CPU when there is none).

    What the collective agreement covers: a missing library, a failed ncclGetUniqueId, an id that did not arrive, and an
    ncclCommInitRank that RETURNS an error on some rank -- every rank then raises, none is left behind in a collective.  What
    it cannot cover: an ncclCommInitRank that never returns on the healthy ranks because a peer died inside it; RCCL's own
    bootstrap time-out (NCCL_SOCKET / comm-init time-outs) is the only way out of that, as for torch's own communicators."""

    _FLOAT32 = 7            # ncclFloat32 (rccl.h)

    def __init__(self, group=None, lib_path=None, device=None):
        import ctypes
        import os
        assert dist.is_initialized(), 'RcclComm needs an initialised torch.distributed group'
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._comm = None
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        dev = torch.device(device)
        # Every step that can fail on SOME ranks only (library not found, bootstrap refused) is followed by an agreement
        # over the torch group -- all_reduce(MIN) of a success flag -- so that either every rank ends up with the
        # communicator or every rank raises: a rank that fell back to c10d on its own while the others sit in
        # ncclCommInitRank / ncclAllGather would hang the job.
        lib = None
        cands = (lib_path,) if lib_path else (os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), 'librccl.so',
                                              'librccl.so.1')
        for cand in cands:
            try:
                lib = ctypes.CDLL(cand)
                break
            except OSError:
                continue
        self._agree(lib is not None, dev, 'librccl.so not found (looked next to torch and on the loader path)')
        self._lib = lib

        class UniqueId(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_ubyte * 128)]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        uid = UniqueId()
        rc = lib.ncclGetUniqueId(ctypes.byref(uid)) if self.rank == 0 else 0
        self._agree(rc == 0, dev, 'ncclGetUniqueId failed on rank 0')
        box = torch.tensor(list(uid.internal), dtype=torch.uint8, device=dev)       # zeros on the other ranks
        dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(box.cpu().tolist()), 128)
        # once more BEFORE the collective init: every rank holds an id (an all-zero one means the broadcast did not deliver)
        self._agree(any(uid.internal), dev, 'the unique id did not arrive')
        comm = ctypes.c_void_p()
        rc = lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank)      # collective over the group
        if rc == 0:
            self._comm = comm
        try:
            self._agree(rc == 0, dev, 'ncclCommInitRank failed'
                        + ('' if rc == 0 else ': ' + lib.ncclGetErrorString(rc).decode()))
        except RuntimeError:
            self.close()
            raise
        self.device = dev
        RcclComm._live.add(self)

    # communicators not yet destroyed (ObjectExchange.close / close_all before destroy_process_group).  Weak references: a
    # communicator whose owner was dropped without close() is destroyed by __del__ instead of living as long as the process.
    import weakref as _weakref
    _live = _weakref.WeakSet()

    def __del__(self):
        # A dropped owner: destroy the communicator -- unless this garbage collection happens to run inside a hipGraph capture,
        # where close()'s device synchronisation would invalidate the capture (and a destroy on one rank while its peers may still be
        # inside a collective is no better): then the communicator is leaked with a warning.  close() / close_all() on every rank
        # before destroy_process_group() remain the supported teardown.
        try:
            if getattr(self, '_comm', None) is None:
                return
            capturing = False
            try:
                capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            except Exception:
                capturing = False
            if capturing:
                import warnings
                warnings.warn('RcclComm dropped during a stream capture: its communicator is leaked (call close() on every rank)',
                              ResourceWarning)
                RcclComm._live.discard(self)
                return
            self.close()
        except Exception:       # interpreter shutdown: modules may be gone
            pass

    def _agree(self, ok, dev, what):
        """Collective: raise on EVERY rank if `ok` is false on ANY rank."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag) == 0:
            raise RuntimeError(what if not ok else f'direct RCCL communicator refused on another rank ({what.split(":")[0]})')

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what} failed: {self._lib.ncclGetErrorString(rc).decode()}')

    def all_gather(self, recv, send):
        """recv (world * n,) <- every rank's send (n,), fp32 contiguous device tensors; on the current stream."""
        assert send.dtype == torch.float32 and recv.dtype == torch.float32 and send.is_contiguous() and recv.is_contiguous()
        assert recv.numel() == self.world * send.numel()
        stream = torch.cuda.current_stream(send.device).cuda_stream if send.is_cuda else 0
        self._check(self._lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), self._FLOAT32, self._comm,
                                            stream), 'ncclAllGather')

    def close(self):
        """ncclCommDestroy (idempotent).  Call on every rank before dist.destroy_process_group(): a communicator that is
        still alive at interpreter exit is torn down by RCCL's own atexit handlers in an order that can hang or warn with
        several ranks."""
        if getattr(self, '_comm', None) is not None:
            if torch.cuda.is_available() and self.device.type == 'cuda':
                torch.cuda.synchronize()            # no collective of ours may still be in flight
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None
        RcclComm._live.discard(self)

    @property
    def closed(self):
        return self._comm is None

    @classmethod
    def close_all(cls):
        """Destroy every live communicator of this process (bench.py / training scripts: before destroy_process_group)."""
        for c in list(cls._live):
            c.close()


class ObjectExchange:
    """The Det step's whole exchange as ONE collective in stream order.

    `start(local, scalars)` packs a few per-rank scalars (the detection loss's `norm_factor` input,
    `monte_carlo_pose_loss.py:53` of EPro-PnP-Det) and this rank's per-object outputs (object axis first, equally padded
    chunk) into one send buffer with ONE kernel and issues ONE `all_gather_into_tensor` on the caller's stream, right after
    the forward that produced the outputs.  `world_mean()` (device tensor: mean over ranks of the scalars) and `objects()`
    (the full `(num_obj, ...)` tensor) read the receive buffer -- stream order is the only synchronisation, the host never
    blocks.  Buffers are allocated once per shape and reused, so the exchange sits inside a hipGraph capture of the step
    like any other launch (RCCL kernels are captured: `bench.py --launch graph`).  A `MonteCarloPoseLoss` accepts the
    exchange in place of its `norm_factor` argument and takes `world_mean()` instead of issuing its own all-reduce.

    Why not a side stream: measured on MI355X (tools/exchange_probe.py, profiles/r03_exchange_probe.txt) a collective on
    the caller's stream adds ~14 us to an eager step and ~2 us to a replayed one; routed over a side stream (fork event,
    c10d's internal stream, join event) it adds 33-48 us eager and 25 us replayed -- every cross-queue dependency is a
    GPU-side bubble of ~10 us here, more than the 10 KB collective itself.
    On device tensors the collective is RCCL's ncclAllGather called directly on the current stream (`RcclComm`: +4 us per
    eager step against +11 us through c10d in the same probe, and no internal stream); `direct=False`, CPU tensors (gloo)
    or a failed communicator set-up take `torch.distributed.all_gather_into_tensor` (`self.route` says which ran).
    The route is agreed on collectively (RcclComm's set-up raises on every rank or on none), so all ranks issue the same
    collective.  `objects()` / `world_mean()` return VIEWS of the reused receive buffer: valid until the next `start()`
    (clone them to keep them longer).  `close()` destroys the direct communicator; call it (or `RcclComm.close_all()`)
    before `dist.destroy_process_group()`.
    Without a process group (or with one rank and `force_collective=False`) no collective is issued."""

    def __init__(self, num_obj, group=None, force_collective=False, direct=True, rccl_lib=None):
        """rccl_lib: path of the RCCL library for the direct route (default: torch's own librccl.so).  With one given, the
        direct route is also taken for CPU tensors -- tests/stubs/rccl_stub.c, the multi-rank test of this branch."""
        self.num_obj, self.group, self.force = int(num_obj), group, bool(force_collective)
        self.direct, self._comm, self.route = bool(direct), None, None
        self._rccl_lib = rccl_lib
        self._key = None
        self._local = self._scal = None
        self.disabled = False       # timing A/B only (bench.py): the step without its exchange

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _active(self):
        return dist.is_initialized() and (self._world() > 1 or self.force) and not self.disabled

    def start(self, local, scalars=None, sum_of=None, sum_scale=1.0, sum_row_weight=None):
        """local (n_local, ...) per-object outputs of this rank's shard; scalars: tensor / float / sequence of them.
        sum_of / sum_scale (HIP tensors): the FIRST scalar is `sum_scale * sum_of.sum()`, evaluated inside the pack launch -- the
        detection head's norm_factor input `(scale * sample_weights[:, None]).sum() / max(2 n, 1)` (deform_pnp_head.py:870) as
        `sum_of=scale, sum_row_weight=sample_weights, sum_scale=1 / max(2 n, 1)`, without its three ATen launches."""
        world = self._world()
        local = local.detach()
        scal = None
        if scalars is not None:
            scal = torch.as_tensor(scalars, dtype=local.dtype, device=local.device).detach().reshape(-1)
        from . import _hip
        fused_pack = local.dtype == torch.float32 and _hip.on_hip_path(local) and (sum_of is None or _hip.on_hip_path(sum_of))
        if sum_of is not None and not self._active():
            if fused_pack:          # no collective, but still one launch for the scalar (same arithmetic as with one)
                from . import functional as F
                n = max(1, 0 if scal is None else scal.numel())
                if getattr(self, '_solo', None) is None or self._solo.numel() != n or self._solo.device != local.device:
                    self._solo = local.new_empty(n)
                F.exchange_pack(self._solo, local[:0], scal, sum_of.detach(), sum_scale, sum_row_weight)
                scal = self._solo
            else:
                first = (self._weighted(sum_of, sum_row_weight).sum() * sum_scale).reshape(1).to(local.dtype)
                scal = first if scal is None else torch.cat((first, scal[1:]))
            sum_of = None
        elif sum_of is not None and not fused_pack:
            first = (self._weighted(sum_of, sum_row_weight).sum() * sum_scale).reshape(1).to(local.dtype)
            scal = first if scal is None else torch.cat((first, scal[1:]))
            sum_of = None
        self._local, self._scal = local, scal
        self._n_scal = (0 if scal is None else scal.numel()) if sum_of is None else max(1, 0 if scal is None else scal.numel())
        if not self._active():
            return self
        chunk = (self.num_obj + world - 1) // world
        row = int(torch.Size(local.shape[1:]).numel())
        key = (world, chunk, tuple(local.shape[1:]), self._n_scal, local.dtype, local.device)
        if key != self._key:
            self._key = key
            self._send = local.new_zeros(self._n_scal + chunk * row)          # [scalars | rows | zero padding]
            self._recv = local.new_empty(world * (self._n_scal + chunk * row))
        self._chunk, self._row = chunk, row
        n = local.shape[0] * row
        if fused_pack:
            from . import functional as F
            F.exchange_pack(self._send, local, scal, None if sum_of is None else sum_of.detach(), sum_scale, sum_row_weight)     # one kernel
        else:
            parts = ([scal] if scal is not None else []) + [local.reshape(-1)]
            torch.cat(parts, out=self._send[:self._n_scal + n])               # one kernel
        direct_ok = local.is_cuda or self._rccl_lib is not None
        if self.direct and direct_ok and self._comm is None:
            try:
                self._comm = RcclComm(self.group, lib_path=self._rccl_lib, device=local.device)
            except Exception as e:         # no librccl / bootstrap refused ON ANY RANK (RcclComm agrees collectively, so
                import warnings            # every rank lands here together): c10d's route still works, say so once
                warnings.warn(f'ObjectExchange: direct RCCL communicator unavailable ({e}); using torch.distributed')
                self.direct = False
        if self.direct and direct_ok:
            self._comm.all_gather(self._recv, self._send)
            self.route = 'rccl ncclAllGather on the current stream'
        else:
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            self.route = f'torch.distributed.all_gather_into_tensor ({dist.get_backend(self.group)})'
        return self

    @staticmethod
    def _weighted(sum_of, row_weight):
        t = sum_of.detach()
        return t if row_weight is None else t * row_weight.detach().reshape((-1,) + (1,) * (t.dim() - 1))

    def close(self):
        """Destroy the direct RCCL communicator, if one was set up (idempotent; the exchange falls back to creating a new one
        on the next start())."""
        if self._comm is not None:
            self._comm.close()
            self._comm = None

    def scalar_slots(self):
        """The FIRST scalar of every rank as a strided (ranks,) view of the receive buffer -- what MonteCarloPoseLoss hands to the
        fused reduce kernel, which averages the ranks itself (no mean launch); the local scalar when no collective ran."""
        assert self._n_scal >= 1, 'start() was called without scalars'
        if not self._active():
            return self._scal[:1]
        return self._recv.view(self._world(), -1)[:, 0]

    def world_mean(self):
        """Mean over ranks of the scalars handed to start(): shape (n_scalars,), or () for a single scalar (a view of the
        receive buffer: valid until the next start())."""
        assert self._n_scal >= 1, 'start() was called without scalars'
        if not self._active():
            return self._scal.reshape(()) if self._n_scal == 1 else self._scal
        m = self._recv.view(self._world(), -1)[:, :self._n_scal].mean(0)
        return m.reshape(()) if self._n_scal == 1 else m

    def objects(self):
        """The gathered (num_obj, ...) per-object outputs, ranks in order, padding trimmed (with an even split a view of the
        receive buffer: valid until the next start())."""
        if not self._active():
            return self._local
        world = self._world()
        per = self._recv.view(world, -1)[:, self._n_scal:]
        shape = (self.num_obj,) + tuple(self._local.shape[1:])
        if self.num_obj == world * self._chunk:          # even split: no padding to trim
            return per.reshape(shape)
        pieces = []
        for r in range(world):
            lo, hi = shard_range(self.num_obj, r, world)
            pieces.append(per[r, :(hi - lo) * self._row])
        return torch.cat(pieces, 0).view(shape)
