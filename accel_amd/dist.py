"""Multi-GPU side of the Accel path: one process per GPU, clips (key-frame
groups) sharded statically across ranks, and an RCCL gather of per-frame
logits to rank 0 -- the only collective on the path.

A clip is the unit of independence: frame t of a clip needs frame t-1's image
and propagated feature (demo.py:176-181,241-243), every key frame resets the
chain.  So ranks never exchange activations; weights are replicated.

The reference's own multi-GPU inference does the same thing with one Python
thread per GPU and host-side merging (dff_rfcn/function/test_rcnn.py:62-82).
"""
import os

import numpy as np


def shard_clips(num_clips, world_size, rank, root_relief=0):
    """Contiguous static partition (config 4: 64 clips -> 8 per GPU).

    root_relief = r > 0: rank 0 -- the root of the logits gather, which besides its own frames receives everybody else's --
    takes r clips fewer than an even share and the least loaded peers absorb them, still contiguous, disjoint and
    complete.  With world_size 1 there is nobody to relieve."""
    if world_size <= 1 or root_relief <= 0:
        per = (num_clips + world_size - 1) // world_size
        lo = min(rank * per, num_clips)
        return list(range(lo, min(lo + per, num_clips)))
    base = num_clips // world_size
    sizes = [base + (1 if r < num_clips % world_size else 0) for r in range(world_size)]
    give = min(int(root_relief), sizes[0])
    sizes[0] -= give
    for _ in range(give):       # to the least loaded peer (the last one among equals)
        r = min(range(world_size - 1, 0, -1), key=lambda k: sizes[k])
        sizes[r] += 1
    lo = sum(sizes[:rank])
    return list(range(lo, lo + sizes[rank]))


def assign_videos_greedy(frame_counts, world_size):
    """test_rcnn.py:62-68: each video goes to the device with the fewest frames so far."""
    loads = [0] * world_size
    owner = []
    for n in frame_counts:
        r = int(np.argmin(loads))
        owner.append(r)
        loads[r] += n
    return owner, loads


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


class _DevPtr(object):
    """Zero-copy torch view of an HBM buffer owned by libaccel_hip."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def as_torch(ptr, shape, dtype="f4", device=0):
    import torch
    return torch.as_tensor(_DevPtr(ptr, shape, {"f4": "<f4", "u1": "|u1"}[dtype]), device="cuda:%d" % device)


def _vote_all(ok, ctx, group):
    """collective AND of a per-rank flag over the group (every rank calls it)"""
    import torch
    import torch.distributed as dist
    if dist.get_world_size(group) == 1:
        return bool(ok)
    dev = "cuda:%d" % ctx.device_id if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


def _vote_any(flag, ctx, group):
    """collective OR of a per-rank flag over the group (every rank calls it)"""
    return not _vote_all(not flag, ctx, group)


def make_comm(ctx, group=None):
    """RCCL communicator of the C ABI (accel_comm_create) spanning the ranks of a torch.distributed group: the 128-byte
    unique id is made on the group's first rank and handed to the others through the process group (the rendezvous
    torch.distributed already did); everything after that is libaccel_hip + librccl, no torch on the data path.

    The decision is COLLECTIVE, in three rounds that every rank takes part in:
      1. every rank probes librccl itself (accel_comm_available: dlopen + the entry points, no communicator) and the ranks
         vote -- ncclCommInitRank blocks until ALL ranks have arrived, so no rank may enter it unless every rank can;
      2. the first rank makes the id and broadcasts it (an exception there travels as the payload);
      3. every rank creates its communicator and the ranks vote again; a partially created set is destroyed.
    Either all ranks get a communicator or all raise AccelError with the reason (FrameGather then falls back to the
    torch.distributed transport on every rank alike)."""
    import torch.distributed as dist
    from . import runtime
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    src = dist.get_global_rank(group, 0) if (group is not None and hasattr(dist, "get_global_rank")) else 0
    try:
        mine = runtime.Comm.available()
    except Exception as e:      # the library itself is missing
        mine = str(e)
    if not _vote_all(mine is None, ctx, group):
        raise runtime.AccelError("C-ABI communicator unavailable on every rank: %s"
                                 % (("rank %d: %s" % (rank, mine)) if mine else "another rank cannot resolve librccl"))
    box = [None]
    if rank == 0:
        try:
            box = [runtime.Comm.unique_id()]
        except Exception as e:
            box = [RuntimeError("rank 0: %s" % (e,))]
    if world > 1:
        dist.broadcast_object_list(box, src=src, group=group)
    comm, why = None, ""
    if isinstance(box[0], Exception) or box[0] is None:
        why = str(box[0])       # the same on every rank: nobody enters ncclCommInitRank
    else:
        try:
            comm = runtime.Comm(ctx, rank, world, box[0])
        except Exception as e:
            why = "rank %d: %s" % (rank, e)
    if not _vote_all(comm is not None, ctx, group) and comm is not None:
        comm.close()
        comm, why = None, "another rank could not create its communicator"
    if comm is None:
        raise runtime.AccelError("C-ABI communicator unavailable on every rank: %s" % (why or "unknown reason"))
    return comm


class FrameGather(object):
    """Asynchronous gather of each frame's output (logits fp32 19xHxW, or the
    uint8 label map) to rank 0, overlapped with the next frame's compute.

    transport "cabi" (default on a GPU): accel_gather_logits of include/accel_hip.h -- the model's output buffer is
    copied into one of two staging slots in compute-stream order (so the next frame may overwrite the output), RCCL
    send/recv to the root run on the library's communication stream beside the next frame; each peer uses its own
    direct xGMI link to the root, no ring, no all-reduce.
    transport "torch": the same protocol through torch.distributed.gather(async_op=True) (fallback when librccl
    cannot be resolved by the library; also the gloo/CPU path of the tests)."""

    def __init__(self, model, ctx, what, shape, dtype, device, group=None, backend_device="cuda", transport="auto", own_bytes=None):
        """shape: what a PEER contributes per frame (the slot size at the root); own_bytes: what THIS rank contributes when that is
        less (only the root may: accel_gather_frames; C-ABI transport only)"""
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.model, self.ctx, self.what = model, ctx, what
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group
        tdt = {"f4": torch.float32, "u1": torch.uint8}[dtype]
        dev = "cuda:%d" % device if backend_device == "cuda" else "cpu"
        self.on_cuda = backend_device == "cuda"
        self.nbytes = int(np.prod(shape)) * (4 if dtype == "f4" else 1)
        self.own_bytes = self.nbytes if own_bytes is None else int(own_bytes)
        if self.own_bytes != self.nbytes and (self.rank != 0 or transport == "torch"):
            raise ValueError("only the root of the C-ABI transport may contribute less than a full slot")
        if self.own_bytes != self.nbytes:
            transport = "cabi"
        self.n = 0
        self.comm, self.transport_note = None, ""
        if self.on_cuda and transport in ("auto", "cabi"):
            # The transport is decided by ALL ranks together.  make_comm succeeds or fails on every rank alike; what a rank does with a
            # failure must be the same everywhere too: a root that contributes less than a full slot (or any rank asked for "cabi")
            # cannot fall back, and then NOBODY may -- a peer that quietly went on to torch.distributed.gather would issue a collective
            # the root never joins (round-4 advisor finding: hang or mismatched collectives with --root-relief).
            must = _vote_any(transport == "cabi", ctx, group)
            try:
                self.comm = make_comm(ctx, group)
            except Exception as e:          # librccl not resolvable from the library: same protocol through torch -- on every rank, or on none
                if must:
                    raise type(e)("%s (a rank of the group needs the C-ABI transport: no rank falls back)" % (e,))
                self.transport_note = "C-ABI communicator unavailable (%s); torch.distributed transport" % (e,)
        self.transport = "cabi" if self.comm is not None else "torch"
        self.recv = [[torch.empty(shape, dtype=tdt, device=dev) for _ in range(self.world)] for _ in range(2)] \
            if self.rank == 0 else [None, None]
        if self.comm is not None:
            # the root receives rank r's block at recv + r*nbytes: one contiguous tensor per slot, `recv` are views of it
            self._flat = [torch.empty((self.world,) + tuple(shape), dtype=tdt, device=dev) for _ in range(2)] if self.rank == 0 else [None, None]
            if self.rank == 0:
                self.recv = [[self._flat[s][r] for r in range(self.world)] for s in range(2)]
            self._src_ptr, _ = model.buffer(what)
            return
        self.stage = [torch.empty(shape, dtype=tdt, device=dev) for _ in range(2)]
        self.work = [None, None]
        self.stream = torch.cuda.ExternalStream(ctx.stream, device=dev) if self.on_cuda else None

    def submit(self):
        s = self.n & 1
        if self.comm is not None:
            self.comm.gather(self._src_ptr, self._flat[s].data_ptr() if self.rank == 0 else None, self.nbytes, 0, self.own_bytes)
        elif self.on_cuda:
            with self.torch.cuda.stream(self.stream):
                # stream-level wait (not a host block): the COMPUTE stream must not refill this staging
                # slot before the gather issued from it two frames ago has read it
                if self.work[s] is not None:
                    self.work[s].wait()
                self.model.read_device(self.what, self.stage[s].data_ptr(), self.nbytes)
                self.work[s] = self.dist.gather(self.stage[s], self.recv[s], dst=0, group=self.group, async_op=True)
        else:   # gloo / CPU path (tests)
            if self.work[s] is not None:
                self.work[s].wait()
            self.stage[s].copy_(self.torch.from_numpy(self.model.read(self.what, tuple(self.stage[s].shape),
                                                                      np.float32 if self.stage[s].dtype == self.torch.float32 else np.uint8)))
            self.work[s] = self.dist.gather(self.stage[s], self.recv[s], dst=0, group=self.group, async_op=True)
        self.n += 1
        return s

    def drain(self):
        if self.comm is not None:
            self.comm.sync()
            return
        for s in (0, 1):
            if self.work[s] is not None:
                if self.on_cuda:
                    with self.torch.cuda.stream(self.stream):
                        self.work[s].wait()
                else:
                    self.work[s].wait()
                self.work[s] = None
        if self.on_cuda:
            self.stream.synchronize()

    def last(self, slot):
        """Root only: list (one per rank) of the tensors gathered in `slot`."""
        return self.recv[slot]

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


class ScoreGather(object):
    """The per-frame gather at SCORE resolution (accel_gather_scores of include/accel_hip.h): every rank sends the fused
    score maps its plans leave in the model's `scores` buffer (B x H/16 x W/16 x 20 fp32: 0.66 MB per 1024x2048 frame instead of the
    159 MB of fp32 logits), the root expands every block into logits + labels with the launch its own plans end with -- bit-identical to
    what the peer computed -- on the communication stream, beside the next frame's compute.  What each peer's xGMI link carries drops
    256x; what remains on the root is the HBM write of the expanded logits (the same bytes a logits gather would have landed there).

    Needs uniform upsampling filters (the reference freezes them: accel_18.py:153): `available(model)` says whether the model has a
    `scores` buffer; use FrameGather("logits") otherwise.  On a GPU the C-ABI transport is the only one (collective decision as in
    FrameGather); backend_device="cpu" is the gloo path of the tests: scores through torch.distributed.gather, expansion through
    model.expand_scores_host."""

    @staticmethod
    def available(model):
        return bool(getattr(model, "has_buffer", lambda n: False)("scores"))

    def __init__(self, model, ctx, B, H, W, device, group=None, own_images=None, ncls=19, backend_device="cuda", emulate_peers=0):
        """emulate_peers = k (one-GPU measurements, world size 1): the root also expands k more blocks per frame -- its own maps again,
        into the image slots k peers would fill -- so that the expansion work of an N = k + 1 job runs beside its compute"""
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.emulate = int(emulate_peers)
        self.model, self.ctx, self.group = model, ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.B, self.own = int(B), int(B if own_images is None else own_images)
        if self.own != self.B and self.rank != 0:
            raise ValueError("only the root may contribute fewer images than a slot holds")
        self.on_cuda = backend_device == "cuda"
        self.map_shape = (H // 16, W // 16, (ncls + 3) // 4 * 4)
        self.n, self.comm = 0, None
        dev = "cuda:%d" % device if self.on_cuda else "cpu"
        self.transport = "cabi" if self.on_cuda else "torch"
        root = self.rank == 0
        self.recv = [torch.empty((self.world, self.B) + self.map_shape, dtype=torch.float32, device=dev) for _ in range(2)] if root else [None, None]
        if self.on_cuda:
            _vote_any(True, ctx, group)      # (the same two collective rounds as FrameGather: every rank requires the C-ABI transport)
            self.comm = make_comm(ctx, group)
            nimg = (self.world + self.emulate) * self.B
            self.logits = torch.empty((nimg, ncls, H, W), dtype=torch.float32, device=dev) if root else None
            self.labels = torch.empty((nimg, H, W), dtype=torch.uint8, device=dev) if root else None
            return
        self.stage = [torch.empty((self.B,) + self.map_shape, dtype=torch.float32) for _ in range(2)]
        self.work = [None, None]
        self.logits = self.labels = None

    def submit(self, plan=None):
        """plan: the runtime.Plan this rank has just run (GPU path; every rank the plan of the same role -- a key plan's map expands
        without the correction bias, a non-key plan's with it)"""
        s = self.n & 1
        if self.comm is not None:
            r = self.rank == 0
            self.comm.gather_scores(plan, self.own, self.B, self.recv[s].data_ptr() if r else None,
                                    self.logits.data_ptr() if r else None, self.labels.data_ptr() if r else None, 0)
            for k in range(self.emulate if r else 0):
                at = (self.world + k) * self.B
                plan.expand_scores(self.recv[s].data_ptr(), self.B, self.logits[at].data_ptr(), self.labels[at].data_ptr(), comm=self.comm)
        else:
            if self.work[s] is not None:
                self.work[s].wait()
            self.stage[s].copy_(self.torch.from_numpy(self.model.read("scores", tuple(self.stage[s].shape), np.float32)))
            self.work[s] = self.dist.gather(self.stage[s], [self.recv[s][r] for r in range(self.world)] if self.rank == 0 else None,
                                            dst=0, group=self.group, async_op=True)
        self.n += 1
        return s

    def expanded(self, slot):
        """Root, CPU path: (logits, labels) of every rank's block of `slot`, expanded by the model (GPU path: self.logits / self.labels
        after drain())"""
        self.work[slot].wait()
        return [self.model.expand_scores_host(self.recv[slot][r].numpy()) for r in range(self.world)]

    def drain(self):
        if self.comm is not None:
            self.comm.sync()
            return
        for s in (0, 1):
            if self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


class Watchdog(object):
    """Fail fast, and say who.  A job of N ranks whose collective one rank never joins does not fail: it hangs -- ncclSend / ncclRecv
    of the C-ABI gathers carry no timeout (torch.distributed's watchdog only sees torch's own collectives).  This object gives every
    PHASE of such a job a deadline: `with wd.phase("preflight gather", 60): ...` marks this rank as having reached the phase in the
    rendezvous store, and a background thread ends the PROCESS (exit code 3, os._exit: the blocked call cannot be interrupted) when the
    phase is still open at its deadline -- after printing, from the store, which ranks never reached it and which did not leave it.
    Under torch.distributed.run one rank exiting ends the others.  `beat()` pushes the deadline of the open phase out again (one call
    per step of a long loop).  store=None (a single process): the deadline alone."""

    EXIT_CODE = 3

    def __init__(self, rank, world, store=None, stream=None, _exit=None):
        import threading
        self.rank, self.world, self.store = int(rank), int(world), store
        self.stream = stream
        self._exit = _exit or os._exit
        self._lock = threading.Lock()
        self._name, self._deadline, self._seconds = None, None, 0.0
        self._stop = False
        self._thread = threading.Thread(target=self._watch, name="accel-watchdog", daemon=True)
        self._thread.start()

    def _key(self, name, what):
        return "accel_watchdog/%s/%s" % (name.replace(" ", "_"), what)

    def _mark(self, name, what):
        if self.store is not None:
            try:
                self.store.set(self._key(name, "%s/%d" % (what, self.rank)), "1")
            except Exception:
                pass

    def _ranks(self, name, what):
        """ranks that have set the marker `what` of phase `name` (None: the store cannot be asked)"""
        if self.store is None:
            return None
        try:
            return [r for r in range(self.world) if self.store.check([self._key(name, "%s/%d" % (what, r))])]
        except Exception:
            return None

    def phase(self, name, seconds):
        wd = self

        class _Phase(object):
            def __enter__(self_):
                import time
                wd._mark(name, "reached")
                with wd._lock:
                    wd._name, wd._seconds, wd._deadline = name, float(seconds), time.monotonic() + float(seconds)
                return wd

            def __exit__(self_, *exc):
                with wd._lock:
                    wd._name, wd._deadline = None, None
                if exc[0] is None:
                    wd._mark(name, "left")
                return False
        return _Phase()

    def beat(self):
        import time
        with self._lock:
            if self._deadline is not None:
                self._deadline = time.monotonic() + self._seconds

    def report(self, name, seconds):
        reached, left = self._ranks(name, "reached"), self._ranks(name, "left")
        msg = "accel watchdog, rank %d of %d: phase '%s' made no progress for %g s" % (self.rank, self.world, name, seconds)
        if reached is not None:
            absent = [r for r in range(self.world) if r not in reached]
            inside = [r for r in reached if r not in (left or [])]
            msg += "; ranks that never reached it: %s; ranks still inside it: %s" % (absent or "none", inside or "none")
        return msg

    def _watch(self):
        import sys
        import time
        while not self._stop:
            time.sleep(0.25)
            with self._lock:
                name, deadline, seconds = self._name, self._deadline, self._seconds
            if deadline is not None and time.monotonic() > deadline:
                out = self.stream or sys.stderr
                try:
                    out.write(self.report(name, seconds) + " -- ending this rank (exit code %d)\n" % self.EXIT_CODE)
                    out.flush()
                finally:
                    self._exit(self.EXIT_CODE)
                return

    def close(self):
        self._stop = True


def default_store():
    """the rendezvous store of the default process group (what torch.distributed.run's ranks met through), or None"""
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        return _get_default_store()
    except Exception:
        return None
