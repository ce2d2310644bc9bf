"""Data-parallel training: one process per GPU, batch sharded, gradients summed with RCCL over xGMI.

The reference's only parallelism is single-process nn.DataParallel (src/models/__init__.py:142-145),
which re-replicates 645 M parameters on every forward. Here every rank owns a full replica whose
parameters and gradients live in ONE flat float32 bucket each (models/convolutional.py), so the
gradient exchange is a handful of large all-reduces over slices of that bucket -- no per-parameter
hooks, no bucket copies.

xGMI is point-to-point (7 links per GPU): RCCL's ring all-reduce of the 2.58 GB gradient bucket is
bound by one link, so the bucket is cut into chunks issued back to back on the RCCL stream; chunk k's
reduction overlaps the fused Adam update of chunk k-1 (each chunk's update only needs its own
reduced slice). The division by world_size is folded into the Adam kernel (`grad_scale`).

Works with the `gloo` backend on CPU tensors for tests (world_size 2).
"""
import os

import torch
import torch.distributed as dist


def force_exchange():
    """SEI_FORCE_EXCHANGE=1: build and run the whole N > 1 machinery -- process group, FlatGradientReducer, sharded
    FlatAdam, early release -- at WORLD_SIZE = 1 too, with none of the single-process short cuts, so that
    `reduce_scatter_tensor` / `all_gather_into_tensor` / the side-stream waits really go through the backend (RCCL on a GPU
    box: the one-GPU rehearsal of the 8-GPU step; tests/dist1_worker.py, bench.py's `secondary.dist1`)."""
    return os.environ.get("SEI_FORCE_EXCHANGE") == "1"


def exchange_active():
    """True when gradient collectives are to be issued: several ranks, or one rank with SEI_FORCE_EXCHANGE=1."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or force_exchange())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_exchange()) and not dist.is_initialized():
        if backend is None:
            # SEI_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL refuses that)
            backend = os.environ.get("SEI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def chunk_bounds(numel, chunk_elems):
    """[(start, stop)] covering [0, numel) in chunks of at most chunk_elems (last one shorter)."""
    if chunk_elems <= 0:
        raise ValueError("chunk_elems must be positive")
    return [(s, min(numel, s + chunk_elems)) for s in range(0, numel, chunk_elems)]


class FlatGradientReducer:
    """Sum a flat gradient bucket across ranks, chunk by chunk, asynchronously.

    reduce_async() issues every chunk's all-reduce; wait(k) blocks the CURRENT stream on chunk k only,
    which lets the optimizer consume chunks in order while later ones are still on the wire.
    """

    def __init__(self, flat_grads, chunk_mib=256, group=None, comm_dtype=torch.float32, early_range=None,
                 mode="all_reduce"):
        """mode: "all_reduce" (one dist.all_reduce per chunk) or "rs_ag" (a reduce-scatter of the chunk into this
        rank's 1/world share, then an all-gather of the shares: the two halves of an all-reduce as separate
        collectives, each of which moves 1/world of the chunk per peer -- on a fully connected xGMI node all 7
        links carry one share each; chunks whose length is not a multiple of world_size fall
        back to all_reduce) or "sharded" (what optim.FlatAdam turns "rs_ag" into: only the reduce-scatter half runs
        here; the optimizer steps this rank's share of every chunk and all-gathers the UPDATED weights instead of the
        gradients -- `shard`, `own_slice`, `gather` below).

        comm_dtype=torch.bfloat16 compresses the exchanged gradients (half the xGMI bytes): the f32
        bucket is cast once into `self.comm`, which is what gets summed and what the optimizer reads.

        early_range=(start, stop): a slice of the bucket whose gradients are final before the backward pass
        ends (graphs.GraphedLossStep.early_grads); chunks never straddle its ends, and reduce_async(early=event)
        exchanges it from a side stream as soon as `event` fires, under the rest of the backward."""
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"unknown gradient exchange mode {mode!r}")
        self.flat = flat_grads
        self.group = group
        self.mode = mode
        self._shards = {}
        self.comm_dtype = comm_dtype
        self.comm = flat_grads if comm_dtype == flat_grads.dtype else torch.empty_like(flat_grads, dtype=comm_dtype)
        self._chunk = max(1, (chunk_mib << 20) // self.comm.element_size())
        self._side = torch.cuda.Stream(device=flat_grads.device) if flat_grads.is_cuda else None
        self.direct_ranges = []       # slices of the bucket that the producer writes into `comm` itself (see below)
        self._direct_now = False
        self._splits = []             # extra positions no chunk may straddle (set_splits)
        self._gather_work = []
        self._plan_lock = None        # why the chunk plan may not change any more (lock_plan), or None
        self._warned_fallback = False
        self.set_early_range(early_range)

    def set_direct_ranges(self, ranges):
        """[(start, stop)]: with a compressed exchange, gradients that the backward pass writes as bf16 straight into
        `self.comm` (graphs.GraphedLossStep: the weight-gradient GEMMs of the deep levels). A step that did so calls
        reduce_async(direct=True), and the cast from the float32 bucket skips those slices."""
        if ranges and self.comm is self.flat:
            raise ValueError("direct bf16 gradients need a compressed (bf16) exchange buffer")
        self.direct_ranges = sorted((int(a), int(b)) for a, b in ranges or [])

    def set_splits(self, positions):
        """Positions of the bucket that no chunk may straddle (optim.FlatAdam: where the weights that the next forward
        pass reads from their bf16 copy begin -- chunks on either side all-gather different buffers). Re-plans."""
        n = self.flat.numel()
        self._splits = sorted({int(p) for p in positions if 0 < int(p) < n})
        self.set_early_range(self.early_range)

    def lock_plan(self, reason):
        """optim.FlatAdam after a sharded step: which slice of the bucket this rank owns follows from the chunk plan, and
        its Adam moments / float32 masters are current for exactly those slices -- a re-plan would silently hand ranks
        slices with stale state. Until unlock_plan() (FlatAdam.consolidate(moments=True)) a CHANGED plan is refused."""
        self._plan_lock = str(reason)

    def unlock_plan(self):
        self._plan_lock = None

    def _share_quantum(self):
        """Shares of a reduce-scattered chunk start on 16-byte boundaries of every buffer: chunk lengths that are multiples
        of 4 x world cut into aligned shares."""
        if self.mode in ("sharded", "rs_ag") and dist.is_initialized():
            return 4 * dist.get_world_size(self.group)
        return 1

    def set_early_range(self, early_range):
        """(Re)plan the chunks; see __init__. Call before the first reduce_async of a step."""
        n = self.flat.numel()
        new_early = None
        cuts = {0, n} | set(self._splits)
        if early_range is not None and 0 <= early_range[0] < early_range[1] <= n:
            lo, hi = int(early_range[0]), int(early_range[1])
            new_early = (lo, hi)
            cuts |= {lo, hi}
        cuts = sorted(cuts)
        parts = list(zip(cuts, cuts[1:]))
        # in the reduce-scatter modes every chunk is a whole number of aligned shares (world sizes such as 3, 5, 6, 7
        # do not divide the power-of-two default): full chunks are rounded down, a part's tail is cut into its largest
        # such piece + a remainder of fewer than 4 x world elements that is all-reduced
        q = self._share_quantum()
        chunk = max(q, self._chunk // q * q)
        bounds, is_early = [], []
        for a, b in parts:
            pieces = []
            for s, e in (chunk_bounds(b - a, chunk) if b > a else []):
                main = (e - s) // q * q
                if q > 1 and 0 < main < e - s:
                    pieces += [(s, s + main), (s + main, e)]
                else:
                    pieces.append((s, e))
            for s, e in pieces:
                bounds.append((a + s, a + e))
                is_early.append(new_early is not None and new_early[0] <= a and b <= new_early[1])
        if self._plan_lock is not None and bounds != getattr(self, "bounds", None):
            raise RuntimeError(f"FlatGradientReducer: the chunk plan may not change now ({self._plan_lock})")
        self.early_range, self.bounds, self._is_early = new_early, bounds, is_early
        # the order in which chunks complete (and the optimizer should consume them): early ones first
        self.order = [k for k, f in enumerate(self._is_early) if f] + [k for k, f in enumerate(self._is_early) if not f]
        self._work = [None] * len(self.bounds)

    def _cast(self, s, e):
        if self.comm is self.flat:
            return
        pos = s
        for a, b in (self.direct_ranges if self._direct_now else []):      # already bf16 in comm
            a, b = max(a, s), min(b, e)
            if a >= b:
                continue
            if a > pos:
                self._cast_run(pos, a)
            pos = max(pos, b)
        if pos < e:
            self._cast_run(pos, e)

    def _cast_run(self, s, e):
        if self.flat.is_cuda and s % 4 == 0:                # sei_cast_bf16 moves aligned quads
            import _native as N
            N.call("sei_cast_bf16", self.flat[s:e].data_ptr(), self.comm[s:e].data_ptr(), e - s)
        else:
            self.comm[s:e].copy_(self.flat[s:e])

    # -- sharded optimizer step (mode "sharded") ---------------------------------------------------------------
    def is_sharded(self, k):
        """Chunk k is reduce-scattered: this rank holds the reduced gradient of `own_slice(k)` only. Shares start on
        16-byte boundaries of every buffer (the fused Adam kernel moves aligned quads), hence the factor 4."""
        s, e = self.bounds[k]
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        return self.mode == "sharded" and exchange_active() and (e - s) % (4 * world) == 0 and s % 4 == 0

    def own_slice(self, k):
        s, e = self.bounds[k]
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n = (e - s) // world
        return s + rank * n, s + (rank + 1) * n

    def shard(self, k):
        """The reduced gradient of own_slice(k) (valid after wait(k))."""
        return self._shards[k]

    def gather(self, k, buffers):
        """All-gather `own_slice(k)` of every buffer in `buffers` (flat tensors over the whole bucket: the updated
        parameters, their bf16 copy) into the chunk's slice on every rank, asynchronously; `wait_gathers()` orders the
        current stream behind them."""
        s, e = self.bounds[k]
        lo, hi = self.own_slice(k)
        in_place = self._in_place()
        for buf in buffers:
            # RCCL: the in-place form (send buffer = this rank's slot of the receive buffer); gloo has none: a 1/world copy
            src = buf[lo:hi] if in_place else buf[lo:hi].clone()
            self._gather_work.append(dist.all_gather_into_tensor(buf[s:e], src, group=self.group, async_op=True))

    def gathered_gradient(self):
        """Collective, tests only: the complete reduced bucket (a copy) after reduce_async in mode "sharded", where
        each rank holds its shares alone."""
        self.wait_all()
        full = self.comm.clone()
        for k in range(len(self.bounds)):
            if self.is_sharded(k):
                s, e = self.bounds[k]
                dist.all_gather_into_tensor(full[s:e], self._shards[k].clone(), group=self.group)
        return full

    def wait_gathers(self):
        for w in self._gather_work:
            w.wait()
        self._gather_work = []

    def _in_place(self):
        """RCCL / NCCL take the in-place forms of both halves -- the reduce-scatter's output is this rank's slot of its
        input, the all-gather's input is this rank's slot of its output (recvbuff == sendbuff + rank * count): no share
        buffers, no 1/world staging copies, and with one rank (secondary.dist1) no copy at all. gloo has no such form.

        OPT-IN (SEI_EXCHANGE_IN_PLACE=1): RCCL has run this code at world size 1 only, where in-place does nothing; the
        default is the out-of-place form (separate share buffers, a 1/world staging copy per gather) that the gloo
        world-2 tests execute. tests/test_ddp_gpu.py::test_in_place_exchange_matches_out_of_place_rccl compares the two
        on a box with >= 2 GPUs. In place, `comm` OUTSIDE own_slice(k) is UNDEFINED after reduce_async in mode "sharded":
        it still holds this rank's unreduced local gradients next to the reduced share, and shard(k) aliases the bucket
        -- read the whole reduced bucket through gathered_gradient() only."""
        return (dist.is_initialized() and dist.get_backend(self.group) == "nccl"
                and os.environ.get("SEI_EXCHANGE_IN_PLACE") == "1")

    def _share_buffer(self, k, chunk, world):
        """Where chunk k's reduce-scatter delivers this rank's share: its own slot of the chunk (in place) or a buffer."""
        n = chunk.numel() // world
        if self._in_place():
            r = dist.get_rank(self.group)
            share = self._shards[k] = chunk[r * n:(r + 1) * n]
            return share
        share = self._shards.get(k)
        if share is None or share.numel() != n or share.dtype != chunk.dtype or share._base is not None:
            share = self._shards[k] = torch.empty(n, dtype=chunk.dtype, device=chunk.device)
        return share

    def _exchange(self, k):
        s, e = self.bounds[k]
        chunk = self.comm[s:e]
        world = dist.get_world_size(self.group)
        if self.is_sharded(k):
            share = self._share_buffer(k, chunk, world)
            self._work[k] = dist.reduce_scatter_tensor(share, chunk, op=dist.ReduceOp.SUM, group=self.group,
                                                       async_op=True)
            return
        if self.mode == "sharded" and not self._warned_fallback and e - s >= 4 * world:
            self._warned_fallback = True
            import warnings
            warnings.warn(f"FlatGradientReducer: chunk [{s}, {e}) cannot be cut into {world} aligned shares; it is "
                          "all-reduced and every rank steps the whole of it")
        if self.mode == "rs_ag" and (e - s) % world == 0:
            share = self._share_buffer(k, chunk, world)
            rs = dist.reduce_scatter_tensor(share, chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if dist.get_backend(self.group) != "nccl":
                rs.wait()             # gloo runs async work on a thread pool: order the two halves by hand
            # (RCCL: both collectives are enqueued on the process group's stream, in this order)
            self._work[k] = dist.all_gather_into_tensor(chunk, share, group=self.group, async_op=True)
            return
        self._work[k] = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_async(self, early=None, direct=False):
        """Cast (if compressed) and all-reduce every chunk asynchronously. early: an event with .wait(stream)
        after which `early_range` is final -- that range is handled on a side stream waiting only for the event,
        the rest on the current stream (i.e. after everything enqueued so far). direct: this step's backward pass
        wrote `direct_ranges` into the exchange buffer itself."""
        self._direct_now = bool(direct) and bool(self.direct_ranges)
        self._work = [None] * len(self.bounds)
        single = not exchange_active()
        if early is not None and self.early_range is not None and self._side is not None:
            lo, hi = self.early_range
            with torch.cuda.stream(self._side):
                early.wait(self._side)
                self._cast(lo, hi)
                if not single:
                    for k in self.order:
                        if self._is_early[k]:
                            self._exchange(k)
            for a, b in ((0, lo), (hi, self.flat.numel())):
                if b > a:
                    self._cast(a, b)
            if single:                                    # the consumer reads self.comm on the current stream
                torch.cuda.current_stream().wait_stream(self._side)
                return
            for k in self.order:
                if not self._is_early[k]:
                    self._exchange(k)
            return
        self._cast(0, self.flat.numel())
        if single:
            return
        for k in self.order:          # the same collective sequence on every rank, whichever path a rank's step took
            self._exchange(k)

    def wait(self, k):
        if self._work[k] is not None:
            self._work[k].wait()

    def wait_all(self):
        for w in self._work:
            if w is not None:
                w.wait()
        self._work = [None] * len(self.bounds)


def broadcast_parameters(flat_params, src=0):
    """Make every replica start from rank `src`'s weights."""
    if exchange_active():
        dist.broadcast(flat_params, src=src)


def all_reduce_mean_scalar(t):
    """Average a scalar tensor over ranks (epoch-level loss logging; not on the per-step path)."""
    if exchange_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t
