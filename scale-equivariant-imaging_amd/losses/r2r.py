"""Recorrupted-to-Recorrupted losses (reference: src/losses/r2r.py, in-tree).

    R2RLoss:    pert = eta * n,  y+ = y + alpha pert,  y- = y - pert / alpha,  metric(A(model(y+)), y-)
    R2REILoss:  R2RLoss(eta = sigma, alpha = 0.5)  +  the EI term with consistent input noise:
                x1 = model(y + 0.5 sigma n1);  x2 = T(x1) (under no_grad if set);  y2 = A(x2);
                x3 = model(y2 + 1.5 sigma n2);  metric(x3, x2)
The three normal draws per call come from torch's generator in the reference's order (pert, n1, n2); tests may
inject them through `_noise=(pert_unit, n1, n2)`.
"""
import torch
from torch.nn import Module

from .ei import mse


class R2RLoss(Module):
    def __init__(self, metric=None, eta=0.1, alpha=0.5):
        super().__init__()
        self.name = "r2r"
        self.metric = metric if metric is not None else mse()
        self.eta = eta
        self.alpha = alpha

    def forward(self, y, physics, model, _unit_noise=None, **kwargs):
        unit = torch.randn_like(y) if _unit_noise is None else _unit_noise
        pert = unit * self.eta
        y_plus = y + pert * self.alpha
        y_minus = y - pert / self.alpha
        output = model(y_plus.contiguous(), physics)
        return self.metric(physics.A(output), y_minus.contiguous())


class R2REILoss(Module):
    def __init__(self, transform, sigma, no_grad=True, metric=None):
        super().__init__()
        self.T = transform
        self.sigma = sigma
        self.no_grad = no_grad
        self.metric = metric if metric is not None else mse()
        self.r2r_loss = R2RLoss(eta=self.sigma, alpha=0.5)

    def forward(self, *kargs, _noise=None, **kwargs):
        n0, n1, n2 = _noise if _noise is not None else (None, None, None)
        return self.r2r_loss(*kargs, _unit_noise=n0, **kwargs) + self.ei_loss(*kargs, _n1=n1, _n2=n2, **kwargs)

    def ei_loss(self, y, physics, model, _n1=None, _n2=None, **kwargs):
        epsilon1 = 0.5 * self.sigma * (torch.randn_like(y) if _n1 is None else _n1)
        x1 = model((y + epsilon1).contiguous(), physics)
        if self.no_grad:
            with torch.no_grad():
                x2 = self.T(x1)
        else:
            x2 = self.T(x1)
        y2 = physics.A(x2)
        epsilon2 = 1.5 * self.sigma * (torch.randn_like(y2) if _n2 is None else _n2)
        x3 = model((y2 + epsilon2).contiguous(), physics)
        return self.metric(x3, x2)
