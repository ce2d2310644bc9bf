"""SURE loss for Gaussian noise with a Monte-Carlo divergence (reference: src/losses/sure.py).

    loss = mean_int((A(x_net) - y)^2) + 2 sigma^2 * mean_int(b * (A(f(y + tau b)) - A(x_net)) / tau) - cst

`int` = the interior [m:-m, m:-m] of every image (m = `margin`; the divergence uses it only with
`cropped_div`), b ~ N(0,1) on that interior and 0 outside, cst = sigma^2 (averaged_cst) or
sigma^2 / batch_size. The two interior reductions and both gradients come out of one fused HIP kernel
(`sei_sure_terms`); the probe y + tau*b is `sei_axpy`.
"""
import torch
import torch.nn as nn

import _native as N
from physics._ops import axpy


class _SureTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, y1, y2, b, margin_div, margin_mse, tau, sigma2, cst):
        """y2 None: `y1` is [A(x_net); A(f(y + tau b))] as ONE tensor of 2B images (ProposedLoss evaluates the operator on both
        model outputs at once): its halves are read in place and one gradient tensor comes back for it -- autograd then has
        no slices to differentiate (two zero fills, two copies and an add per step). The loss value
        c_mse * mse_sum + c_div * div_sum - cst is formed by the kernel's own last stage (sei_sure_loss)."""
        joint = y2 is None
        B, C, H, W = y.shape
        for name, t in (("y", y), ("y1", y1), ("b", b)) + (() if joint else (("y2", y2),)):
            N.check_tensor(t, name)
            if t.shape != ((2 * B, C, H, W) if joint and name == "y1" else y.shape):
                raise ValueError("SURE: y, A(x_net), A(f(y + tau b)) and b must share one shape")
        n_div = B * C * (H - 2 * margin_div) * (W - 2 * margin_div)
        n_mse = B * C * (H - 2 * margin_mse) * (W - 2 * margin_mse)
        c_mse, c_div = 1.0 / n_mse, 2.0 * sigma2 / n_div
        out = torch.empty(3, dtype=torch.float32, device=y.device)
        work = torch.empty(2 * N.SEI_REDUCE_BLOCKS, dtype=torch.float32, device=y.device)
        if joint:
            g = torch.empty_like(y1)
            half = y.numel() * y.element_size()
            ptrs = (y1.data_ptr(), y1.data_ptr() + half, g.data_ptr(), g.data_ptr() + half)
            ctx.save_for_backward(g)
        else:
            g1, g2 = torch.empty_like(y), torch.empty_like(y)
            ptrs = (y1.data_ptr(), y2.data_ptr(), g1.data_ptr(), g2.data_ptr())
            ctx.save_for_backward(g1, g2)
        ctx.joint = joint
        N.call("sei_sure_loss", y.data_ptr(), ptrs[0], ptrs[1], b.data_ptr(), B * C, H, W, margin_div, margin_mse, tau,
               c_mse, c_div, float(cst), out.data_ptr(), ptrs[2], ptrs[3], work.data_ptr())
        return out[2]

    @staticmethod
    def backward(ctx, go):
        if ctx.joint:
            (g,) = ctx.saved_tensors
            return (None, N.scale_by(g, go)) + (None,) * 7
        g1, g2 = ctx.saved_tensors
        return (None, N.scale_by(g1, go), N.scale_by(g2, go)) + (None,) * 6


def draw_probe(y, margin):
    """b: N(0,1) on the interior, 0 on the `margin`-wide border (reference mc_div, :9-22)."""
    if margin == 0:
        return torch.randn_like(y)
    b = torch.zeros_like(y)
    b[:, :, margin:-margin, margin:-margin] = torch.randn(
        y.size(0), y.size(1), y.size(2) - 2 * margin, y.size(3) - 2 * margin, device=y.device, dtype=y.dtype)
    return b


def embed_probe(y, b_interior, margin):
    """Zero-extend an interior-shaped draw to y's shape (for injected randomness in tests)."""
    if margin == 0:
        return b_interior.contiguous()
    b = torch.zeros_like(y)
    b[:, :, margin:-margin, margin:-margin] = b_interior
    return b


class SureGaussianLoss(nn.Module):
    def __init__(self, sigma, tau=1e-2, margin=0, cropped_div=False, averaged_cst=False):
        super().__init__()
        self.name = "SureGaussian"
        self.sigma2 = sigma**2
        self.tau = tau
        assert margin is not None
        self.margin = margin
        self.cropped_div = cropped_div
        self.averaged_cst = averaged_cst

    @property
    def div_margin(self):
        return self.margin if self.cropped_div else 0

    def forward(self, y, x_net, physics, model, b=None, y1=None, y2=None, y12=None, **kwargs):
        """`b` (full-size probe), `y1` = A(x_net) and `y2` = A(model(y + tau b)) may be supplied by a
        caller that has already evaluated them (ProposedLoss batches the two network passes: `y12` = both as one
        tensor of 2B images)."""
        y = y.contiguous()
        if b is None:
            b = draw_probe(y, self.div_margin)
        cst = self.sigma2 if self.averaged_cst else self.sigma2 / y.size(0)
        if y12 is not None:
            loss = _SureTerms.apply(y, y12.contiguous(), None, b, self.div_margin, self.margin, self.tau, self.sigma2, cst)
        else:
            if y1 is None:
                y1 = physics.A(x_net)
            if y2 is None:
                y2 = physics.A(model(axpy(y, b, self.tau)))
            loss = _SureTerms.apply(y, y1.contiguous(), y2.contiguous(), b, self.div_margin, self.margin, self.tau,
                                    self.sigma2, cst)
        from os import environ
        if "_TEMPORARY_HOTFIX" in environ:       # reference :68-74
            assert physics.rate is not None
            return physics.rate**2 * loss
        return loss
