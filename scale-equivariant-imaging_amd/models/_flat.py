"""One contiguous float32 bucket for all parameters of a model, a matching gradient bucket and a bf16 copy.

Shared by the two backbones (ConvolutionalModel, SwinIR). After construction or a device move the parameters are
re-homed into `flat_params`, with `flat_grads` beside it: backward kernels accumulate straight into views of the
gradient bucket, the fused Adam is one launch over the bucket (optim.FlatAdam) and the RCCL exchange is a few large
collectives over slices of it (parallel.FlatGradientReducer). `state_dict()` / `load_state_dict()` are untouched:
the parameters keep their names and shapes, only their storage moves.
"""
import torch

from . import _ops


class FlatParameterBucket:
    """Mixin for an nn.Module (list it BEFORE Module in the bases). Call `_init_bucket()` at the end of __init__."""

    def _init_bucket(self):
        self.flat_params = None
        self.flat_grads = None
        self.flat_shadow = None
        self.flat_shadow_only_start = None
        self.compute_dtype = None                 # "f32" / "bf16": this model's own mode; None = models._ops' process default
        self._sei_plain_state = _ops._new_plain_state()
        self._sei_zero_ranges = None
        # load_state_dict copies into the parameters: THIS model's cached bf16 shadows are stale afterwards (and every
        # float32 master is what the file said: nothing of a sharded optimizer step's staleness survives a load)
        self.register_load_state_dict_post_hook(FlatParameterBucket._after_load)

    @staticmethod
    def _after_load(module, incompatible):
        _ops.set_stale_masters(module, None)
        _ops.weights_updated(module)

    @staticmethod
    def _goes_last(p):
        """GEMM weights (1x1 convolutions) go last in the bucket, so that everything else -- the part of the
        gradient bucket that must be zeroed every step in store mode -- is one contiguous head; and of those, the ones
        that the bf16 throughput mode reads ONLY through their bf16 copy (both extents multiples of 32: every block
        that owns such a weight takes the bf16 path -- the gate is the block's INPUT channel count,
        models/_ops.use_bf16_blocks: shape[1] for conv2 / Upsample / Downsample, shape[0] for conv3, hence both) come last
        of all: with a sharded optimizer step (optim.FlatAdam under several GPUs) only that copy of them is all-gathered.
        0 = head, 1 = 1x1 weights read as float32 by some path, 2 = bf16-copy-only weights."""
        if not (p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1):
            return 0
        return 2 if p.shape[0] % 32 == 0 and p.shape[1] % 32 == 0 else 1

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.flatten_parameters()
        return out

    def flatten_parameters(self):
        """Re-home every parameter into one contiguous bucket (and prepare the gradient bucket)."""
        params = list(self.parameters())
        if not params:
            return
        params.sort(key=self._goes_last)
        dev, dt = params[0].device, params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in params):
            return
        # every parameter starts on a 256-byte boundary of the bucket (64 float32): 16-byte vector access
        # for all of them whatever the sizes before; the padding stays zero in all three buckets
        align = 64
        offsets, total = [], 0
        self.flat_shadow_only_start = None          # bucket offset where the bf16-copy-only weights begin
        for p in params:
            if self.flat_shadow_only_start is None and self._goes_last(p) == 2:
                self.flat_shadow_only_start = total
            offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        if self.flat_shadow_only_start is None:
            self.flat_shadow_only_start = total
        flat = torch.zeros(total, dtype=dt, device=dev)
        grads = torch.zeros(total, dtype=dt, device=dev)
        for p, off in zip(params, offsets):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p._sei_grad_view = grads[off:off + n].view(p.shape)
            p.grad = None
        self.flat_params, self.flat_grads = flat, grads
        _ops.register_gradient_range(self, grads)       # backward functions find this model's bookkeeping by address
        # bf16 copy of the whole bucket for the throughput mode (written by the fused Adam kernel)
        self.flat_shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        # validity of the bf16 bucket is tracked per model: {"gen": generation it was written for,
        # "version": torch version counter of each parameter at that time}
        self._sei_plain_state = _ops._new_plain_state()
        self._sei_zero_ranges = None
        for p, off in zip(params, offsets):
            p._sei_bucket_offset = off
        if self.flat_shadow is not None:
            for p, off in zip(params, offsets):
                p._sei_shadow_view = self.flat_shadow[off:off + p.numel()]
                p._sei_plain_state = self._sei_plain_state
                p._sei_shadow = None

    def zero_grad_flat(self, store_weight_grads=False):
        """One memset for the whole model; leaves every p.grad attached to the bucket.

        store_weight_grads=True (GraphedLossStep, after `plan_weight_grad_store`): the 1x1-convolution weight
        gradients are not zeroed -- the step's first launch into each of them stores (models/_ops.py)."""
        ranges = self._sei_zero_ranges if store_weight_grads else None
        if ranges is None:
            self.flat_grads.zero_()
        elif self.flat_grads.is_cuda:
            import ctypes
            import _native as N
            for k in range(0, len(ranges), 8):                   # one launch per 8 gaps (three or four gaps at defaults)
                part = ranges[k:k + 8]
                pairs = (ctypes.c_ulonglong * (2 * len(part)))(*[v for off_n in part for v in off_n])
                N.call("sei_zero_ranges", self.flat_grads.data_ptr(), ctypes.cast(pairs, ctypes.c_void_p), len(part))
        else:
            for off, n in ranges:
                self.flat_grads[off:off + n].zero_()
        _ops.begin_step(store=ranges is not None, store_min=getattr(self, "_sei_store_min", 0), owner=self)
        for p in self.parameters():
            p.grad = p._sei_grad_view

    def plan_weight_grad_store(self, min_numel=1 << 20):
        """After at least one eager step: the parts of the gradient bucket that still need zeroing when the
        weight gradients recorded by models/_ops.py are stored rather than accumulated. Only gradients of at least
        `min_numel` elements are stored: a small one costs nothing to zero with its neighbours, while its GEMM is a
        split-K launch whose STORING form needs a zero-fill launch of its own in front."""
        base, esz = self.flat_grads.data_ptr(), self.flat_grads.element_size()
        total = self.flat_grads.numel()
        self._sei_store_min = int(min_numel)
        skip = sorted(((ptr - base) // esz, n) for ptr, n in _ops.weight_grad_views(owner=self).items()
                      if base <= ptr < base + total * esz and n >= min_numel)
        ranges, pos = [], 0
        for off, n in skip:
            if off > pos:
                ranges.append((pos, off - pos))
            pos = max(pos, off + n)
        if pos < total:
            ranges.append((pos, total - pos))
        self._sei_zero_ranges = ranges if skip else None
        return self._sei_zero_ranges
