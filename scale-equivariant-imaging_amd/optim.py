"""Fused Adam over the model's flat parameter bucket (replaces torch.optim.Adam at demo/train.py:157-186
for the U-Net; same update rule, one kernel launch per chunk instead of a foreach sweep per tensor).

Keeps a torch.optim.Optimizer-compatible surface (param_groups with "lr", state_dict / load_state_dict,
zero_grad, step) so the reference's schedulers and checkpoint code drive it unchanged.
"""
import torch

import _native as N


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, reducer=None):
        backbone = model.get_backbone() if hasattr(model, "get_backbone") else model
        if getattr(backbone, "flat_params", None) is None:
            raise ValueError("FlatAdam needs a model whose parameters live in one flat bucket")
        self.backbone = backbone
        self.reducer = reducer
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__([backbone.flat_params], defaults)
        flat = backbone.flat_params
        self.state[flat] = {"step": 0, "exp_avg": torch.zeros_like(flat), "exp_avg_sq": torch.zeros_like(flat)}

    def zero_grad(self, set_to_none=True):
        self.backbone.zero_grad_flat()

    @torch.no_grad()
    def step(self, closure=None):
        flat, grads = self.backbone.flat_params, self.backbone.flat_grads
        N.check_tensor(flat, "flat_params")
        group = self.param_groups[0]
        st = self.state[flat]
        st["step"] += 1
        b1, b2 = group["betas"]
        world = 1
        bounds = [(0, flat.numel())]
        grads_16 = False
        if self.reducer is not None:
            from parallel import world_size
            world = world_size()
            bounds = self.reducer.bounds
            grads = self.reducer.comm                     # the (possibly bf16-compressed) reduced bucket
            grads_16 = grads.dtype == torch.bfloat16
        from models import _ops
        shadow = getattr(self.backbone, "flat_shadow", None) if _ops.get_compute_dtype() == "bf16" else None
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
