"""Blur physics (reference: src/physics/blur/__init__.py).

`BlurV2` (the default, --physics_v2) and the legacy `Blur` (--no-physics_v2) compute the same
circular convolution in the reference (FFT route :205-223 vs padded conv2d loops :38-77, agreeing to
5e-7); both are served here by the same LDS-tiled HIP kernel, `A_adjoint` by its transposed form.
"""
import torch

from ._base import LinearPhysics
from ._ops import CircularBlurOp, apply_linear


class BlurV2(LinearPhysics):
    def __init__(self, kernel):
        super().__init__()
        self.kernel = kernel
        self.filter = self.kernel        # alias read by get_loss / demo/test.py (reference :202)
        self._op = CircularBlurOp(kernel)

    def A(self, x):
        return apply_linear(self._op, x.contiguous())

    def A_adjoint(self, y):
        return apply_linear(self._op, y.contiguous(), transpose=True)


class Blur(LinearPhysics):
    """Legacy surface: Blur(filter, padding='circular', device). Only circular padding is on the
    hot path (src/physics/__init__.py:45); other paddings are rejected loudly."""

    def __init__(self, filter, padding="circular", device="cpu", **kwargs):
        super().__init__()
        if padding != "circular":
            raise ValueError(f"Blur: only padding='circular' is supported by this build, got {padding!r}")
        self.padding = padding
        self.device = device
        self.filter = torch.nn.Parameter(filter, requires_grad=False).to(device)
        self._op = CircularBlurOp(filter)

    def A(self, x):
        return apply_linear(self._op, x.contiguous())

    def A_adjoint(self, y):
        return apply_linear(self._op, y.contiguous(), transpose=True)
