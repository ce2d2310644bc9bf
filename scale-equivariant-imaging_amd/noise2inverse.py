"""Noise2Inverse evaluation glue (reference call surface: src/noise2inverse.py, after Hendriksen et al.).

The measurement is split into `num_splits` interleaved row sets; each set is back-projected on its own (FFT inverse
filter for deblurring, the physics' pseudo-inverse otherwise) and the backbone sees sums of back-projections:
with the "X:1" strategy every input is the sum of all but one slice and the left-out slice is the target.
`demo/test.py --noise2inverse` averages nothing: it SUMS the backbone's outputs over the four inputs (:82-86).

Not on the training hot path (the reference's TrainingDataset stores its `noise2inverse` flag and never reads it,
src/datasets/__init__.py:63-90): plain torch ops on whatever device the measurement lives on, torch.fft for the
inverse filter exactly as upstream -- with a Gaussian blur the filter divides by transfer values near 1e-9, so its
output is defined by the FFT's own rounding and only the same FFT reproduces it (tests pin it on the CPU against
tests/golden/g12_noise2inverse.npz, generated from the reference).
"""
from itertools import combinations

import numpy as np
import torch
from torch.nn import Module


def _splits(num_splits, strategy):
    """(input index tuples, target index sets) in the order upstream enumerates them (:88-90, :126-129)."""
    every = set(range(num_splits))
    inputs = list(combinations(every, num_splits - 1 if strategy == "X:1" else 1))
    return inputs, [every - set(group) for group in inputs]


class InverseFilter(Module):
    """x = irfft2(rfft2(y) / rfft2(psf)) with the kernel embedded at the origin of an image-sized PSF (:49-70)."""

    def __init__(self, kernel):
        super().__init__()
        self.kernel = kernel

    def forward(self, y):
        assert y.dim() == 4
        H, W = y.shape[-2:]
        kh, kw = self.kernel.shape[-2:]
        psf = y.new_zeros((H, W))
        psf[:kh, :kw] = self.kernel
        psf = torch.roll(psf, (-(kh // 2), -(kw // 2)), dims=(-2, -1))
        spectrum = torch.fft.rfft2(y, dim=(-2, -1))
        otf = torch.fft.rfft2(psf, dim=(-2, -1)).expand(spectrum.shape)
        return torch.fft.irfft2(spectrum / otf, dim=(-2, -1), s=(H, W))


class ImageSlices(Module):
    def __init__(self, num_splits, task, physics_filter, degradation_inverse_fn):
        super().__init__()
        self.num_splits = num_splits
        if task == "deblurring":
            assert physics_filter is not None and physics_filter.dim() == 4
            self.backproject = InverseFilter(kernel=physics_filter.squeeze(0).squeeze(0))
        else:
            self.backproject = degradation_inverse_fn

    def measurement_slices(self, y):
        rows = torch.arange(y.shape[-2], device=y.device) % self.num_splits
        return [y * (rows == j).to(y.dtype).view(1, 1, -1, 1) for j in range(self.num_splits)]

    def forward(self, y):
        return [self.backproject(part) for part in self.measurement_slices(y)]


class Noise2InverseModel(Module):
    def __init__(self, backbone, task, physics_filter, degradation_inverse_fn, num_splits=4, strategy="X:1"):
        super().__init__()
        self.backbone = backbone
        self.num_splits = num_splits
        self.strategy = strategy
        self.transform = ImageSlices(num_splits=num_splits, task=task, physics_filter=physics_filter,
                                     degradation_inverse_fn=degradation_inverse_fn)

    def compute_inputs(self, y):
        parts = self.transform(y)
        groups, _ = _splits(self.num_splits, self.strategy)
        return [torch.stack([parts[j] for j in group]).sum(dim=0) for group in groups]

    def forward(self, y):
        return torch.stack([self.backbone(v) for v in self.compute_inputs(y)]).sum(dim=0)


class Noise2InverseTransform(Module):
    """(x, y) -> (target, input): one of the splits, chosen by numpy's global generator (:131)."""

    def __init__(self, task, physics_filter, degradation_inverse_fn, strategy="X:1", num_splits=4):
        super().__init__()
        self.strategy = strategy
        self.num_splits = num_splits
        self.transform = ImageSlices(num_splits=num_splits, task=task, physics_filter=physics_filter,
                                     degradation_inverse_fn=degradation_inverse_fn)

    def forward(self, x, y):
        parts = self.transform(y)
        groups, targets = _splits(self.num_splits, self.strategy)
        pick = np.random.randint(0, len(groups))
        total = lambda idxs: torch.stack([parts[j] for j in idxs]).sum(dim=0)
        return total(targets[pick]), total(groups[pick])
