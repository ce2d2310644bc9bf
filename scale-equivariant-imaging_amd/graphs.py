"""hipGraph capture of the training step's device work.

One proposed-loss step is ~620 kernel launches of 5-500 us; launched eagerly from Python the GPU idles
between them. `GraphedLossStep` captures zero_grad + loss forward + backward ONCE (torch.cuda.graph =
hipGraph on ROCm) over static buffers and replays it per step:

    graphed = GraphedLossStep(loss, model, optimizer, y_shape)
    loss_value = graphed(x, y)          # crop (host RNG, as the reference) -> copy -> replay
    optimizer.step()                    # outside the graph: its bias corrections are host scalars

Inside the graph the 1x1-convolution weight gradients of the step's two model calls are produced by one
GEMM each that STORES its result (models/_ops.py), so only the small remainder of the gradient bucket is
zeroed per step.

What stays outside the graph: the random 48-crop of Loss.forward (two CPU randint draws + a strided
copy into the static input), the step's device-side random draws, the gradient all-reduce and the fused
Adam launch. The draws (probe b, scale rates/centres, measurement noise) are made eagerly before every
replay, by the same calls in the same order as an eager step (`loss.draw`), into static buffers the
captured kernels read: with equal seeds a replayed step and an eager step see the same numbers, so
replay == eager is an equality the tests assert, and no torch RNG kernel sits inside the graph.

Only steps without host-side randomness or host syncs can be captured (`graph_safe` of the method-level
loss: proposed with the padded scaling transform and no antialias, sure, supervised, css); `can_capture`
is the allow-list train.py consults, everything else runs eagerly.
"""
import torch


def can_capture(loss_module):
    """True when `loss_module` (a losses.Loss) cropping + method can be replayed from a hipGraph."""
    return loss_module.crop_fn is not None and bool(getattr(loss_module.loss, "graph_safe", False))


def _copy_nested(dst, src):
    """dst <- src over matching dicts / sequences of tensors (None entries must match)."""
    if isinstance(dst, torch.Tensor):
        dst.copy_(src.reshape(dst.shape))
    elif hasattr(dst, "buffer") and hasattr(src, "buffer") and dst.buffer.shape == src.buffer.shape:
        dst.buffer.copy_(src.buffer)                   # models.swinir.DropMasks: every mask of a model call in one copy
    elif isinstance(dst, dict):
        for key, value in dst.items():
            _copy_nested(value, src[key])
    elif isinstance(dst, (list, tuple)):
        if len(dst) != len(src):
            raise ValueError("random draws changed structure between capture and replay")
        for d, s_ in zip(dst, src):
            _copy_nested(d, s_)
    elif dst is not None or src is not None:
        raise ValueError("random draws changed structure between capture and replay")


class GraphedLossStep:
    def __init__(self, loss_module, model, optimizer, crop_shape, warmup=3, store_weight_grads=True,
                 early_release=False, fuse_optimizer=False, fuse_min_numel=1 << 24, store_min_numel=1 << 20,
                 direct_bf16_grads=True, count_nodes=False):
        """loss_module: a `losses.Loss`; crop_shape: (B, 3, S, S) of the cropped measurement y.
        fuse_optimizer: apply the optimizer step of the stored weight gradients of at least `fuse_min_numel` elements
        inside the GEMM that produces them (optim.FlatAdam.fuse_weight_updates; one GPU, bf16 mode, a loss whose
        model calls all merge into one weight-gradient GEMM per weight).
        direct_bf16_grads: with a reducer whose exchange buffer is bf16, the same weights' gradients are written into
        that buffer as bf16 by their GEMM (`self.direct_views`; the caller passes reduce_async(direct=True) after a
        replayed step).
        count_nodes: keep the captured hipGraph_t long enough to count its nodes (`self.node_counts` = (kernel nodes,
        all nodes); bench.py reports them: a kernel node costs ~1.5 us of launch structure per replay)."""
        self.loss_module = loss_module
        self.inner = loss_module.loss                # method-level loss working on cropped tensors
        self.model = model
        self.optimizer = optimizer
        backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
        if getattr(backbone, "flat_grads", None) is None:
            raise ValueError("GraphedLossStep needs a flattened model")
        self.backbone = backbone
        if not getattr(self.inner, "graph_safe", False):
            raise ValueError(f"{type(self.inner).__name__} in this configuration draws random numbers on the host or "
                             "synchronises with it inside the step: it cannot be captured (graphs.can_capture)")
        device = backbone.flat_params.device
        self.static_y = torch.zeros(crop_shape, dtype=torch.float32, device=device)
        self.static_x = None
        if getattr(self.inner, "needs_x", False):
            r = loss_module.xy_size_ratio if loss_module.crop_fn is not None else 1
            self.static_x = torch.zeros(tuple(crop_shape[:2]) + (crop_shape[2] * r, crop_shape[3] * r),
                                        dtype=torch.float32, device=device)
        # static homes of the step's random numbers, refreshed by `_draw` before every replay
        state = torch.cuda.get_rng_state(device)
        self.static_draws = self.inner.draw(self.static_y, self.model)
        torch.cuda.set_rng_state(state, device)        # sizing the buffers must not advance the generator

        self.store_weight_grads = False

        # the root gradient (autograd would fill a fresh one per step); allocated OUTSIDE the graph's pool, so it must live
        # as long as the graph whose kernels read it
        unit = self._unit_grad = torch.ones((), dtype=torch.float32, device=device)

        def fwd_bwd():
            self.backbone.zero_grad_flat(store_weight_grads=self.store_weight_grads)
            value = self.inner(x=self.static_x, y=self.static_y, model=self.model, draws=self.static_draws)
            if value.dim() == 0 and value.dtype == torch.float32:
                value.backward(unit)
            else:
                value.backward()
            return value.detach()

        from models import _ops
        _ops.weight_grad_views(reset=True, owner=backbone)           # record this model's weight-gradient views only
        _ops.reset_splitk_counters(device)
        # warm-up AND capture on this one stream: the split-K workspace of the GEMMs belongs to a (device, stream), and a
        # stream that first meets it while capturing gets none (models/_ops.py splitk_workspace)
        side = self._capture_stream = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            self._splitk_ws = _ops.own_splitk_workspace(device)      # this graph's own, alive as long as the graph is
            for _ in range(warmup):                  # settle caches / allocator outside capture
                fwd_bwd()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        # The captured step rebuilds the TRANSPOSED bf16 weight shadows from the plain bf16 copy, which
        # the fused Adam refreshes every step; make that copy current now, and re-check before each replay.
        self._ops = _ops
        _ops.refresh_plain_shadow(self.backbone)
        # The warm-up steps showed which gradients are written by the merged weight-gradient GEMMs; the
        # captured step stores those instead of accumulating and zeroes only the rest of the bucket.
        self.store_weight_grads = store_weight_grads and \
            self.backbone.plan_weight_grad_store(store_min_numel) is not None
        # Early release of the largest gradients: an EXTERNAL event recorded inside the captured backward right
        # after the two largest adjacent weight gradients (the bottleneck block: 83 % of the bucket at defaults)
        # have been written; `early_grads` = (event, start, stop) in bucket elements, or None.
        self.early_grads = None
        if self.store_weight_grads and early_release:
            self.early_grads = self._plan_early_release(_ops)
        self.fused_views, self.fused_table = [], None
        if fuse_optimizer and self.store_weight_grads and hasattr(optimizer, "fuse_weight_updates") \
                and _ops.get_compute_dtype(backbone) == "bf16" and getattr(optimizer, "reducer", None) is None:
            grads = self.backbone.flat_grads
            base, esz = grads.data_ptr(), grads.element_size()
            for prm in self.backbone.parameters():
                g = prm._sei_grad_view
                off = (g.data_ptr() - base) // esz
                if prm.dim() == 4 and prm.shape[2:] == (1, 1) and prm.numel() >= fuse_min_numel and off % 4 == 0 \
                        and prm.shape[1] % 8 == 0 and g.data_ptr() in _ops.merged_weight_grads(backbone):
                    self.fused_views.append(g.view(prm.shape[0], prm.shape[1]))
            if self.fused_views:
                self.fused_table = optimizer.fuse_weight_updates(self.fused_views)
                _ops.set_fused_adam(*self.fused_table, owner=backbone)
        # Several GPUs with a bf16-compressed exchange: the same weights' gradients go to the exchange buffer as bf16
        # (no float32 copy, no cast pass); the reducer is told which slices not to cast after a replayed step.
        self.direct_views = []
        reducer = getattr(optimizer, "reducer", None)
        if direct_bf16_grads and self.store_weight_grads and reducer is not None and reducer.comm is not reducer.flat \
                and reducer.comm.dtype == torch.bfloat16 and _ops.get_compute_dtype(backbone) == "bf16":
            grads = self.backbone.flat_grads
            table, ranges = {}, []
            for prm in self.backbone.parameters():
                g = prm._sei_grad_view
                off = (g.data_ptr() - grads.data_ptr()) // grads.element_size()
                if prm.dim() == 4 and prm.shape[2:] == (1, 1) and prm.numel() >= fuse_min_numel and off % 4 == 0 \
                        and prm.shape[1] % 8 == 0 and g.data_ptr() in _ops.merged_weight_grads(backbone):
                    view = g.view(prm.shape[0], prm.shape[1])
                    table[view.data_ptr()] = reducer.comm[off:off + prm.numel()].view(prm.shape[0], prm.shape[1])
                    ranges.append((off, off + prm.numel()))
                    self.direct_views.append(view)
            if table:
                reducer.set_direct_ranges(ranges)
                _ops.set_direct_bf16_grads(table, owner=backbone)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if count_nodes else torch.cuda.CUDAGraph()
        self.node_counts = None
        try:
            with torch.cuda.graph(self.graph, stream=side):
                self.static_loss = fwd_bwd()
            if self.direct_views and _ops.direct_bf16_launches(backbone) != {v.data_ptr() for v in self.direct_views}:
                raise RuntimeError("bf16 gradients into the exchange buffer: not every registered weight was written "
                                   "by the captured step")
            if self.fused_views and _ops.fused_adam_launches(backbone) != {v.data_ptr() for v in self.fused_views}:
                raise RuntimeError("fused optimizer step: not every registered weight was updated by the captured step")
            if count_nodes:
                import _native
                self.node_counts = _native.graph_kernel_nodes(self.graph)
                self.graph.instantiate()
        except Exception:
            if self.fused_views:
                optimizer.unfuse_weight_updates()
            if self.direct_views:
                reducer.set_direct_ranges([])
            raise
        finally:
            _ops.release_splitk_workspace(device, side, self._splitk_ws)
            _ops.set_weight_grad_milestone(None, None, owner=backbone)
            _ops.set_fused_adam(None, None, owner=backbone)
            _ops.set_direct_bf16_grads(None, owner=backbone)
        self.backbone.zero_grad_flat()

    def _plan_early_release(self, _ops):
        base, esz = self.backbone.flat_grads.data_ptr(), self.backbone.flat_grads.element_size()
        total = self.backbone.flat_grads.numel()
        views = sorted(((n, ptr) for ptr, n in _ops.weight_grad_views(owner=self.backbone).items()
                        if base <= ptr < base + total * esz), reverse=True)
        if len(views) < 2:
            return None
        (n0, p0), (n1, p1) = views[0], views[1]
        lo, hi = min(p0, p1), max(p0, p1)
        lo_n = n0 if lo == p0 else n1
        start, stop = (lo - base) // esz, (hi - base) // esz + (n0 if hi == p0 else n1)
        if (hi - lo) // esz - lo_n >= 64 or (stop - start) < total // 4:       # not adjacent, or not worth it
            return None
        import _native
        event = _native.ExternalEvent()
        _ops.set_weight_grad_milestone({p0, p1}, event, owner=self.backbone)
        return (event, int(start), int(stop))

    def _draw(self, given=None):
        """Fresh random numbers for the next replay: the eager step's own draw calls (or `given`, a dict of the
        same tensors), copied into the buffers the graph reads."""
        if self.static_draws is None:
            return
        if given is None and hasattr(self.inner, "draw_into") \
                and self.inner.draw_into(self.static_draws, self.static_y, self.model):
            return                                  # drawn straight into the static buffers (same calls, same order)
        fresh = given if given is not None else self.inner.draw(self.static_y, self.model)
        _copy_nested(self.static_draws, fresh)

    def __call__(self, x, y, draws=None):
        """One replay on the crop of (x, y); draws: inject the step's random numbers (tests) instead of drawing."""
        crop = self.loss_module.crop_fn
        if crop is not None and self.static_x is None and hasattr(crop, "draw_offsets") \
                and tuple(y.shape[:2]) == tuple(self.static_y.shape[:2]) and crop.size == self.static_y.shape[-1]:
            # the loss never reads x: draw the offset exactly as CropPair.forward does and copy y's window straight into
            # the static input -- no zero-padded copies of the two 256 x 256 batches first
            i, j, _, _ = crop.draw_offsets(y.shape)
            crop.write_y(y, i, j, self.static_y)
        else:
            if crop is not None:
                x, y = crop(x, y, xy_size_ratio=self.loss_module.xy_size_ratio)
            if tuple(y.shape) != tuple(self.static_y.shape):
                raise ValueError(f"graphed step was captured for {tuple(self.static_y.shape)}, got {tuple(y.shape)}")
            self.static_y.copy_(y)
            if self.static_x is not None:
                self.static_x.copy_(x)
        self._draw(draws)
        if not self._ops.plain_shadow_is_current(self.backbone):   # weights changed by something other than FlatAdam
            self._ops.refresh_plain_shadow(self.backbone)
        if self.fused_views:
            self.optimizer.prepare_step()          # the step's Adam scalars, read by the fused GEMM epilogues
        self.graph.replay()
        return self.static_loss
