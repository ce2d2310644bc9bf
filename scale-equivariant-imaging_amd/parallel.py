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


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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

    def __init__(self, flat_grads, chunk_mib=256, group=None, comm_dtype=torch.float32):
        """comm_dtype=torch.bfloat16 compresses the exchanged gradients (half the xGMI bytes): the f32
        bucket is cast once into `self.comm`, which is what gets summed and what the optimizer reads."""
        self.flat = flat_grads
        self.group = group
        self.comm_dtype = comm_dtype
        self.comm = flat_grads if comm_dtype == flat_grads.dtype else torch.empty_like(flat_grads, dtype=comm_dtype)
        self.bounds = chunk_bounds(flat_grads.numel(), max(1, (chunk_mib << 20) // self.comm.element_size()))
        self._work = []

    def reduce_async(self):
        self._work = []
        if self.comm is not self.flat:
            if self.flat.is_cuda:
                import _native as N
                N.call("sei_cast_bf16", self.flat.data_ptr(), self.comm.data_ptr(), self.flat.numel())
            else:
                self.comm.copy_(self.flat)
        if world_size() == 1:
            return
        for s, e in self.bounds:
            self._work.append(dist.all_reduce(self.comm[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self, k):
        if self._work:
            self._work[k].wait()

    def wait_all(self):
        for w in self._work:
            w.wait()
        self._work = []


def broadcast_parameters(flat_params, src=0):
    """Make every replica start from rank `src`'s weights."""
    if world_size() > 1:
        dist.broadcast(flat_params, src=src)


def all_reduce_mean_scalar(t):
    """Average a scalar tensor over ranks (epoch-level loss logging; not on the per-step path)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t
