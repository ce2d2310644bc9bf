"""Super-resolution physics (reference: src/physics/downsampling/__init__.py:9-35).

A = antialiased bicubic resize by 1/rate, evaluated as the separable banded matrix product
Wv x Wh^T by `sei_resample_sepband`; the autograd backward and the `true_adjoint` use the transposed
bands, the deprecated adjoint the plain (a=-0.75) bicubic upsampling matrix.
"""
from . import _bands
from ._base import LinearPhysics
from ._ops import SeparableResampleOp, apply_linear


class Downsampling(LinearPhysics):
    def __init__(self, rate, antialias, true_adjoint=False):
        super().__init__()
        self.rate = rate
        self.antialias = antialias
        self.true_adjoint = true_adjoint
        build = _bands.aa_bicubic_matrix if antialias else _bands.plain_bicubic_matrix
        self._down = SeparableResampleOp(lambda n: build(n, 1 / rate), inverse_length=lambda n: n * rate)
        self._up = SeparableResampleOp(lambda n: _bands.plain_bicubic_matrix(n, rate))

    def A(self, x):
        return apply_linear(self._down, x.contiguous())

    def A_adjoint(self, y):
        if self.true_adjoint:
            return apply_linear(self._down, y.contiguous(), transpose=True)
        return apply_linear(self._up, y.contiguous())
