"""Fused Adam over the model's flat parameter bucket (replaces torch.optim.Adam at demo/train.py:157-186
for the U-Net; same update rule, one kernel launch per chunk instead of a foreach sweep per tensor).

Keeps a torch.optim.Optimizer-compatible surface (param_groups with "lr", state_dict / load_state_dict,
zero_grad, step) so the reference's schedulers and checkpoint code drive it unchanged. `state_dict()` is
written, and `load_state_dict()` read, in torch.optim.Adam's own per-parameter layout over
`model.parameters()` (what the reference saves under "optimizer", src/training.py:23-31), so `--RESUME`
interchanges checkpoints with the reference and with `--no-fused_optimizer` runs; the flat moment buckets
are an internal layout that never reaches a file.
"""
import torch

import _native as N


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, reducer=None,
                 shard_step=True):
        """reducer: parallel.FlatGradientReducer under several GPUs. shard_step: with the reducer's "rs_ag" exchange,
        take the optimizer step on this rank's 1 / world share of every chunk only and all-gather the UPDATED weights
        instead of the reduced gradients (`_step_sharded`): the same bytes on the wire, 1 / world of the 30 bytes per
        parameter that Adam moves through HBM (19.4 GB per step for the default U-Net: 2.4 GB per rank at 8 GPUs)."""
        backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
        if getattr(backbone, "flat_params", None) is None:
            raise ValueError("FlatAdam needs a model whose parameters live in one flat bucket")
        self.backbone = backbone
        self.reducer = reducer
        if reducer is not None and reducer.mode == "rs_ag" and shard_step:
            reducer.mode = "sharded"
            start = getattr(backbone, "flat_shadow_only_start", None)
            reducer.set_splits([start] if start is not None else [])
        self._named = list(model.parameters())        # the order torch.optim.Adam(model.parameters()) indexes by
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__([backbone.flat_params], defaults)
        flat = backbone.flat_params
        self.state[flat] = {"step": 0, "exp_avg": torch.zeros_like(flat), "exp_avg_sq": torch.zeros_like(flat)}
        # optimizer step inside the weight-gradient GEMMs (fuse_weight_updates): slices of the bucket stepped there,
        # the step number the device scalars were prepared for, and whether an eager backward pass filled the bucket
        self._fused_ranges = None
        self._prepared_for = None
        self._eager_grads = False
        self._hyper_dev = None

    @property
    def _master_stale(self):
        """Sharded step, bf16 mode: the float32 weights of other ranks' shares are out of date on this rank (kept on the
        backbone, where models/_ops refuses to rebuild bf16 copies from them until `consolidate()`)."""
        return self.backbone._sei_plain_state["stale"] is not None

    def zero_grad(self, set_to_none=True):
        self.backbone.zero_grad_flat()
        self._eager_grads = True          # an eager backward pass follows: its gradients cover the whole bucket

    # -- optimizer step inside the weight-gradient GEMMs (one GPU, bf16 mode, captured step) ----------------------
    def fuse_weight_updates(self, grad_views):
        """grad_views: 2-D views into the flat gradient bucket whose step is to be applied by the GEMM that
        produces them (graphs.GraphedLossStep: the two deepest levels, 98.8 % of the bucket at defaults -- the 4-byte
        gradient round trip and 26 of Adam's 30 bytes per parameter move into that GEMM's epilogue, under its MFMA
        work). `step()` then covers the rest of the bucket; `prepare_step()` must run before every captured step so
        that the device scalars the epilogue reads belong to the step about to be taken. Returns the table for
        models._ops.set_fused_adam."""
        if self.reducer is not None:
            raise ValueError("fused weight updates need the complete gradient on this GPU (no gradient exchange)")
        from models import _ops
        flat, grads = self.backbone.flat_params, self.backbone.flat_grads
        st = self.state[flat]
        shadow = getattr(self.backbone, "flat_shadow", None) if _ops.get_compute_dtype(self.backbone) == "bf16" else None
        table, ranges = {}, []
        for gv in grad_views:
            off = (gv.data_ptr() - grads.data_ptr()) // grads.element_size()
            n = gv.numel()
            if not (0 <= off and off + n <= grads.numel()) or gv.dim() != 2 or not gv.is_contiguous():
                raise ValueError("fused weight updates: not a contiguous 2-D view of the flat gradient bucket")
            if off % 4 or gv.shape[1] % 8 or (flat.data_ptr() | st["exp_avg"].data_ptr() | st["exp_avg_sq"].data_ptr()) % 16:
                raise ValueError("fused weight updates: the epilogue moves aligned quads (bucket offset % 4, columns % 8)")
            pick = lambda buf: buf[off:off + n].view(gv.shape)
            table[gv.data_ptr()] = (pick(flat), pick(st["exp_avg"]), pick(st["exp_avg_sq"]),
                                    None if shadow is None else pick(shadow))
            ranges.append((off, off + n))
        self._fused_ranges = sorted(ranges)
        self._hyper_dev = torch.zeros(6, dtype=torch.float32, device=flat.device)
        self._prepared_for = None
        return table, self._hyper_dev

    def unfuse_weight_updates(self):
        self._fused_ranges = None

    def prepare_step(self):
        """Scalars of the NEXT step (lr of the moment, bias corrections of step + 1) -> the device array read by the
        fused epilogues; a launch on the current stream, i.e. ordered in front of the replay that uses them."""
        if not self._fused_ranges:
            return
        group, st = self.param_groups[0], self.state[self.backbone.flat_params]
        b1, b2 = group["betas"]
        # as launch arguments, not through a host buffer: the host runs steps ahead of the GPU, and a pinned buffer copied
        # asynchronously would already hold a later step's scalars when the copy finally executes
        N.call("sei_adam_scalars_to_device", float(group["lr"]), float(b1), float(b2), float(group["eps"]),
               float(group["weight_decay"]), int(st["step"]) + 1, self._hyper_dev.data_ptr())
        self._prepared_for = int(st["step"]) + 1
        self._eager_grads = False

    def _step_bounds(self, total, whole=False):
        """The bucket minus the ranges whose update a GEMM epilogue has applied."""
        fused = self._fused_ranges
        if not fused or whole:
            return [(0, total)]
        out, pos = [], 0
        for lo, hi in fused:
            if lo > pos:
                out.append((pos, lo))
            pos = max(pos, hi)
        if pos < total:
            out.append((pos, total))
        return out

    # -- checkpoint interchange ----------------------------------------------------------------
    _TORCH_ADAM_GROUP = {"amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                         "differentiable": False, "fused": None, "decoupled_weight_decay": False}

    def _slices(self):
        flat = self.backbone.flat_params
        base, esz = flat.data_ptr(), flat.element_size()
        for p in self._named:
            off = (p.data_ptr() - base) // esz
            if not (0 <= off and off + p.numel() <= flat.numel()):
                raise RuntimeError("a model parameter lives outside the flat bucket (flatten_parameters not run?)")
            yield p, off

    # -- sharded step under several GPUs ------------------------------------------------------------------------
    def _shadow_only(self, start):
        """Chunk starting at `start` holds weights whose only reader in the next forward pass is their bf16 copy."""
        from models import _ops
        first = getattr(self.backbone, "flat_shadow_only_start", None)
        return (_ops.get_compute_dtype(self.backbone) == "bf16" and getattr(self.backbone, "flat_shadow", None) is not None
                and first is not None and start >= first)

    def _step_sharded(self, update, shadow):
        """`update(lo, hi, grads, grads_are_bf16)` applies the step to flat[lo:hi]. Chunks in the order their
        reduce-scatters complete: step this rank's share, then all-gather what the other ranks need of it -- the bf16
        copy alone for the weights that the bf16 mode reads only through it (98.8 % of the default U-Net: 2 bytes per
        parameter on the wire instead of 4, the float32 masters of other ranks' shares go stale until `consolidate`),
        the float32 weights (and their bf16 copy) for everything else. A chunk that cannot be cut into aligned shares
        was all-reduced; every rank steps the whole of it."""
        from models import _ops
        red = self.reducer
        flat = self.backbone.flat_params
        stale = self.backbone._sei_plain_state["stale"]
        for k in red.order:
            s, e = red.bounds[k]
            red.wait(k)
            if not red.is_sharded(k):
                update(s, e, red.comm[s:e], red.comm.dtype == torch.bfloat16)
                continue
            lo, hi = red.own_slice(k)
            share = red.shard(k)
            update(lo, hi, share, share.dtype == torch.bfloat16)
            if self._shadow_only(s):
                red.gather(k, [shadow])
                stale = (s, e) if stale is None else (min(stale[0], s), max(stale[1], e))
            else:
                red.gather(k, [flat] + ([shadow] if shadow is not None else []))
        red.wait_gathers()
        _ops.set_stale_masters(self.backbone, stale)
        # this rank's moments and masters are current for the shares of THIS plan only (parallel.FlatGradientReducer)
        red.lock_plan("a sharded optimizer step has been taken: call optimizer.consolidate() (moments included) on every "
                      "rank before changing the chunk plan")

    def consolidate(self, moments=True):
        """Collective (every rank calls it): bring the float32 weights -- and, for a checkpoint, both Adam moments -- of
        the other ranks' shares up to date on this rank. After a sharded step each rank holds current float32 state for
        its own shares only; `state_dict()`, `model.get_weights()` and anything else that reads the whole bucket in
        float32 needs this first (train.py calls it before every save)."""
        red = self.reducer
        if red is None or red.mode != "sharded":
            return
        st = self.state[self.backbone.flat_params]
        bufs = ([self.backbone.flat_params] if self._master_stale else []) + \
            ([st["exp_avg"], st["exp_avg_sq"]] if moments and st["step"] > 0 else [])
        if bufs:
            for k in range(len(red.bounds)):
                if red.is_sharded(k):
                    stale = self._shadow_only(red.bounds[k][0])
                    red.gather(k, [b for b in bufs if b is not self.backbone.flat_params or stale])
            red.wait_gathers()
        from models import _ops
        _ops.set_stale_masters(self.backbone, None)
        if moments:
            red.unlock_plan()             # every rank holds complete masters and moments: any chunk plan may follow

    def state_dict(self):
        """torch.optim.Adam's layout: state[i] = {step, exp_avg, exp_avg_sq} for parameter i of
        model.parameters(); empty before the first step, as torch's. (Sharded step under several GPUs: call
        `consolidate()` on every rank first.)"""
        st = self.state[self.backbone.flat_params]
        state = {}
        if st["step"] > 0:
            for i, (p, off) in enumerate(self._slices()):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(st["step"])),
                            "exp_avg": st["exp_avg"][off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": st["exp_avg_sq"][off:off + n].view(p.shape).clone()}
        group = dict(self._TORCH_ADAM_GROUP)
        group.update({k: v for k, v in self.param_groups[0].items() if k != "params"})
        group["params"] = list(range(len(self._named)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        count = sum(len(g["params"]) for g in groups)
        if len(groups) != 1 or count != len(self._named):
            raise ValueError(f"optimizer state has {len(groups)} group(s) over {count} parameters; this model has "
                             f"{len(self._named)} parameters in one group (a torch.optim.Adam(model.parameters()) "
                             "state of the same architecture is expected)")
        if any(groups[0].get(k) for k in ("amsgrad", "maximize")):
            raise ValueError("amsgrad / maximize Adam states are not supported by the fused optimizer")
        st = self.state[self.backbone.flat_params]
        saved = state_dict["state"]
        steps = {int(float(e["step"])) for e in saved.values()}
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not a state the fused optimizer can continue")
        st["step"] = steps.pop() if steps else 0
        st["exp_avg"].zero_()
        st["exp_avg_sq"].zero_()
        for i, (p, off) in enumerate(self._slices()):
            e = saved.get(i, saved.get(str(i)))
            if e is None:
                continue
            n = p.numel()
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(e['exp_avg'].shape)} != parameter {tuple(p.shape)}")
            st["exp_avg"][off:off + n].copy_(e["exp_avg"].reshape(-1))
            st["exp_avg_sq"][off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
        self.param_groups[0].update({k: v for k, v in groups[0].items()
                                     if k != "params" and k not in self._TORCH_ADAM_GROUP})

    @torch.no_grad()
    def step(self, closure=None):
        flat, grads = self.backbone.flat_params, self.backbone.flat_grads
        N.check_tensor(flat, "flat_params")
        group = self.param_groups[0]
        st = self.state[flat]
        st["step"] += 1
        whole = False
        if self._fused_ranges and self._prepared_for != st["step"]:
            # not a replayed step: fine after zero_grad() + an eager backward pass (a short last batch), which wrote every
            # gradient the ordinary way; anything else would step the fused weights on gradients nobody computed
            if not self._eager_grads:
                st["step"] -= 1
                raise RuntimeError("FlatAdam.step(): part of this step is applied inside the captured backward pass, "
                                   "which was not replayed with prepare_step() for this step "
                                   "(graphs.GraphedLossStep does both)")
            whole = True
        self._eager_grads = False
        b1, b2 = group["betas"]
        world = 1
        bounds = self._step_bounds(flat.numel(), whole)
        grads_16 = False
        exchanging = False
        if self.reducer is not None:
            from parallel import exchange_active, world_size
            world = world_size()
            exchanging = exchange_active()
            bounds = self.reducer.bounds
            grads = self.reducer.comm                     # the (possibly bf16-compressed) reduced bucket
            grads_16 = grads.dtype == torch.bfloat16
        from models import _ops
        shadow = getattr(self.backbone, "flat_shadow", None) if _ops.get_compute_dtype(self.backbone) == "bf16" else None
        if self.reducer is not None and self.reducer.mode == "sharded" and exchanging:
            def update(lo, hi, g, g16):
                N.call("sei_adam_fused", flat[lo:hi].data_ptr(), g.data_ptr(), int(g16), st["exp_avg"][lo:hi].data_ptr(),
                       st["exp_avg_sq"][lo:hi].data_ptr(), hi - lo, float(group["lr"]), float(b1), float(b2),
                       float(group["eps"]), float(group["weight_decay"]), int(st["step"]), 1.0 / world,
                       None if shadow is None else shadow[lo:hi].data_ptr())
            self._step_sharded(update, shadow)
            _ops.weights_updated(self.backbone, plain_shadow_written=shadow is not None)
            return
        order = self.reducer.order if self.reducer is not None else range(len(bounds))
        for k in order:                                   # chunks in the order their all-reduces complete
            s, e = bounds[k]
            if self.reducer is not None:
                self.reducer.wait(k)
            N.call("sei_adam_fused", flat[s:e].data_ptr(), grads[s:e].data_ptr(), int(grads_16),
                   st["exp_avg"][s:e].data_ptr(),
                   st["exp_avg_sq"][s:e].data_ptr(), e - s, float(group["lr"]), float(b1), float(b2),
                   float(group["eps"]), float(group["weight_decay"]), int(st["step"]), 1.0 / world,
                   None if shadow is None else shadow[s:e].data_ptr())
        # bf16 weight shadows must be rebuilt before the next forward (the plain copy just was, if asked)
        _ops.weights_updated(self.backbone, plain_shadow_written=shadow is not None)
