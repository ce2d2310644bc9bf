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


class LinearPhysics(Module):
    """Protocol: A, A_adjoint, __call__(x) = noise_model(A(x)), attribute noise_model."""

    def __init__(self, **kwargs):
        super().__init__()
        self.noise_model = lambda x: x

    def A(self, x):
        raise NotImplementedError

    def A_adjoint(self, y):
        raise NotImplementedError

    def A_dagger(self, y):
        raise NotImplementedError(
            "A_dagger (pseudo-inverse) is used only by the InverseFilter / Noise2Inverse baselines, "
            "which are outside the training hot path of this build")

    def forward(self, x):
        return self.noise_model(self.A(x))
