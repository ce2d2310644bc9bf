"""Equivariant-imaging and supervised losses (deepinv v0.2.0 `EILoss`, `SupLoss`, `mse`, as the
reference configures them at src/losses/__init__.py:17-22,117-122; restated from their documented
behaviour -- deepinv is not part of the reference tree).

    EI:  x2 = T(x_net) (under no_grad when stop_gradient), y2 = physics(x2) = A(x2) + sigma n,
         x3 = model(y2, physics),  loss = weight * mean((x3 - x2)^2)
"""
import torch
from torch.nn import Module

import _native as N


class _MseTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        N.check_tensor(a, "a")
        N.check_tensor(b, "b")
        if a.shape != b.shape:
            raise ValueError(f"mse: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        n = a.numel()
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        ga = torch.empty_like(a)
        work = torch.empty(N.SEI_REDUCE_BLOCKS, dtype=torch.float32, device=a.device)
        N.call("sei_mse_loss", a.data_ptr(), b.data_ptr(), n, 2.0 * weight / n, weight / n, out.data_ptr(), ga.data_ptr(),
               work.data_ptr())                  # (the weighted mean is formed by the kernel's last stage: no 0-dim torch mul)
        ctx.save_for_backward(ga)
        return out[1]

    @staticmethod
    def backward(ctx, go):
        (ga,) = ctx.saved_tensors
        g = N.scale_by(ga, go)
        return g, (-g if ctx.needs_input_grad[1] else None), None


class mse(Module):
    """Mean squared error over all elements (deepinv.loss.metric.mse)."""

    def forward(self, a, b, weight=1.0):
        return _MseTerms.apply(a.contiguous(), b.contiguous(), float(weight))


class SupLoss(Module):
    def __init__(self, metric=None):
        super().__init__()
        self.name = "supervised"
        self.metric = metric if metric is not None else mse()

    def forward(self, x_net, x, **kwargs):
        return self.metric(x_net, x)


class EILoss(Module):
    def __init__(self, transform, metric=None, apply_noise=True, weight=1.0, no_grad=False):
        super().__init__()
        self.name = "ei"
        self.metric = metric if metric is not None else mse()
        self.weight = weight
        self.T = transform
        self.noise = apply_noise
        self.no_grad = no_grad

    def forward(self, x_net, physics, model, transform_params=None, noise=None, **kwargs):
        """transform_params / noise: the transform's draws and the unit measurement noise, when the caller has
        drawn them already (losses.ProposedLoss.draw); drawn here otherwise, as deepinv does."""
        T = self.T if transform_params is None else (lambda v: self.T(v, params=transform_params))
        if self.no_grad:
            with torch.no_grad():
                x2 = T(x_net)
        else:
            x2 = T(x_net)
        if not self.noise:
            y2 = physics.A(x2)
        elif noise is None:
            y2 = physics(x2)
        else:
            y2 = physics.noise_model(physics.A(x2), noise=noise)
        x3 = model(y2, physics)
        if isinstance(self.metric, mse):
            return self.metric(x3, x2, weight=self.weight)
        return self.weight * self.metric(x3, x2)
