"""Host-side construction of the separable resampling matrices and their band storage.

These are constants of an operator (like the blur taps): computed once in float64 numpy, stored as
float32 band arrays on the device, consumed by `sei_resample_sepband`.

Formulas (restating ATen, as the reference reaches them through F.interpolate):
  * antialiased bicubic (a = -0.5), `_upsample_bicubic2d_aa`: used by Downsampling.A
    (reference src/physics/downsampling/__init__.py:16-19) and the EI antialias pre-filter
    (src/transforms.py:46-57);
  * plain bicubic (a = -0.75, align_corners=False, clamped taps), `upsample_bicubic2d`: the deprecated
    A_adjoint (downsampling/__init__.py:33-34) and the non-antialiased 'normal' transform.
"""
from math import floor

import numpy as np


def _cubic(t, a):
    t = abs(t)
    if t <= 1.0:
        return ((a + 2.0) * t - (a + 3.0)) * t * t + 1.0
    if t < 2.0:
        return (((t - 5.0) * t + 8.0) * t - 4.0) * a
    return 0.0


def resized_length(n_in, scale_factor):
    """Output length of F.interpolate(scale_factor=...): floor(n_in * scale_factor) in double."""
    return int(floor(float(n_in) * float(scale_factor)))


def aa_bicubic_matrix(n_in, scale_factor):
    """Dense (n_out, n_in) antialiased-bicubic matrix for F.interpolate(scale_factor, antialias=True)."""
    n_out = resized_length(n_in, scale_factor)
    scale = 1.0 / float(scale_factor)
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        centre = scale * (o + 0.5)
        first = max(0, int(centre - support + 0.5))
        last = min(n_in, int(centre + support + 0.5))
        w = np.array([_cubic((j - centre + 0.5) * inv, -0.5) for j in range(first, last)])
        m[o, first:last] = w / w.sum()
    return m


def aa_bicubic_matrix_to_size(n_in, n_out):
    """Dense (n_out, n_in) antialiased-bicubic matrix for F.interpolate(size=n_out, antialias=True): as
    aa_bicubic_matrix with the scale ATen derives from the sizes, n_in / n_out (area_pixel_compute_scale with
    align_corners=False and no scale_factor)."""
    scale = float(n_in) / float(n_out)
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        centre = scale * (o + 0.5)
        first = max(0, int(centre - support + 0.5))
        last = min(n_in, int(centre + support + 0.5))
        w = np.array([_cubic((j - centre + 0.5) * inv, -0.5) for j in range(first, last)])
        m[o, first:last] = w / w.sum()
    return m


def plain_bicubic_matrix(n_in, scale_factor):
    """Dense (n_out, n_in) matrix of F.interpolate(scale_factor, mode='bicubic') without antialias."""
    n_out = resized_length(n_in, scale_factor)
    scale = 1.0 / float(scale_factor)
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        src = scale * (o + 0.5) - 0.5
        f = floor(src)
        t = src - f
        co = (_cubic(t + 1.0, -0.75), _cubic(t, -0.75), _cubic(1.0 - t, -0.75), _cubic(2.0 - t, -0.75))
        for d in range(4):
            m[o, min(max(int(f) - 1 + d, 0), n_in - 1)] += co[d]
    return m


def plain_bicubic_matrix_to_size(n_in, n_out):
    """Dense (n_out, n_in) matrix of F.interpolate(size=n_out, mode='bicubic', align_corners=False): as
    plain_bicubic_matrix with the scale ATen derives from the sizes (n_in / n_out)."""
    scale = float(n_in) / float(n_out)
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        src = scale * (o + 0.5) - 0.5
        f = floor(src)
        t = src - f
        co = (_cubic(t + 1.0, -0.75), _cubic(t, -0.75), _cubic(1.0 - t, -0.75), _cubic(2.0 - t, -0.75))
        for d in range(4):
            m[o, min(max(int(f) - 1 + d, 0), n_in - 1)] += co[d]
    return m


def to_band(dense):
    """Dense (n_out, n_in) -> (weights (n_out, nb) f32, lo (n_out,) i32, nb, max step of lo).
    Band starts are made non-decreasing (a row whose leading weights are exact zeros of the cubic
    simply keeps them inside its band)."""
    n_out, n_in = dense.shape
    lo = np.zeros(n_out, dtype=np.int64)
    hi = np.zeros(n_out, dtype=np.int64)
    for o in range(n_out):
        nz = np.nonzero(dense[o])[0]
        lo[o], hi[o] = (nz[0], nz[-1] + 1) if nz.size else (n_in, 0)
    lo = np.minimum.accumulate(lo[::-1])[::-1]          # running min from the end
    hi = np.maximum.accumulate(hi)                      # running max from the start
    lo = np.minimum(lo, n_in - 1)
    hi = np.maximum(hi, lo + 1)
    nb = int((hi - lo).max())
    w = np.zeros((n_out, nb), dtype=np.float32)
    for o in range(n_out):
        w[o, : hi[o] - lo[o]] = dense[o, lo[o]:hi[o]]
    step = int(np.diff(lo).max()) if n_out > 1 else 0
    return w, lo.astype(np.int32), nb, step
