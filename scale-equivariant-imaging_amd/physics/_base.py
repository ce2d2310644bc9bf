"""Minimal stand-ins for the deepinv classes the reference builds its physics on
(deepinv v0.2.0 `LinearPhysics`, `GaussianNoise`; not vendored in the reference, restated from their
documented behaviour -- SURVEY.md a6, 8c)."""
import torch
from torch.nn import Module

from ._ops import axpy


class GaussianNoise(Module):
    """y + sigma * N(0, 1), drawn from the global torch generator of y's device."""

    def __init__(self, sigma=0.1):
        super().__init__()
        self.sigma = sigma

    def forward(self, x, noise=None):
        if noise is None:
            noise = torch.randn_like(x)
        return axpy(x.contiguous(), noise.contiguous(), self.sigma)


def _conjugate_gradient(op, b, max_iter, tol):
    """deepinv.optim.utils.conjugate_gradient (v0.2.0): x from 0, stop when the squared residual falls below tol^2."""
    dot = lambda u, v: (u * v).flatten().sum()
    x = torch.zeros_like(b)
    r = b.clone()
    p = r
    rs = dot(r, r)
    for _ in range(int(max_iter)):
        Ap = op(p.contiguous())
        alpha = rs / dot(p, Ap)
        x = x + p * alpha
        r = r + Ap * (-alpha)
        rs_new = dot(r, r)
        if float(rs_new) < tol ** 2:
            break
        p = r + p * (rs_new / rs)
        rs = rs_new
    return x


class LinearPhysics(Module):
    """Protocol: A, A_adjoint, __call__(x) = noise_model(A(x)), attribute noise_model."""

    def __init__(self, **kwargs):
        super().__init__()
        self.noise_model = lambda x: x

    def A(self, x):
        raise NotImplementedError

    def A_adjoint(self, y):
        raise NotImplementedError

    max_iter, tol = 50, 1e-3        # deepinv v0.2.0 LinearPhysics defaults

    def A_dagger(self, y):
        """Least-squares pseudo-inverse by conjugate gradients on the normal equations, as deepinv v0.2.0's
        LinearPhysics.A_dagger does (restated from its documented behaviour: UNPINNED; used by the `InverseFilter` model
        kind, /root/reference/src/models/__init__.py:22-28, never by the training step): with an over-complete operator
        (fewer unknowns than measurements) solve A^T A x = A^T y, otherwise A A^T u = y and return A^T u. Built on this
        operator's own A / A_adjoint (the HIP kernels); the handful of inner products are torch reductions."""
        Aty = self.A_adjoint(y)
        overcomplete = Aty.numel() < y.numel()
        if overcomplete:
            op, b = (lambda v: self.A_adjoint(self.A(v))), Aty
        else:
            op, b = (lambda v: self.A(self.A_adjoint(v))), y
        x = _conjugate_gradient(op, b, self.max_iter, self.tol)
        return x if overcomplete else self.A_adjoint(x)

    def forward(self, x):
        return self.noise_model(self.A(x))
