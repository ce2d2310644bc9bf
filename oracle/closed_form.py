"""ORACLE (test infrastructure) -- numpy float64 closed forms of the hot-path operators.

These restate each reference operator as the explicit arithmetic the HIP kernels implement
(no FFT, no torch resize/grid_sample calls), so that the kernels' formulas are themselves pinned
against the reference's golden outputs (tests/test_oracle_golden.py). Pure numpy; small inputs.
"""
from math import ceil, floor

import numpy as np

# ----------------------------------------------------------------------------------------------
# blur: BlurV2.A (src/physics/blur/__init__.py:205-223) == direct circular convolution
# ----------------------------------------------------------------------------------------------
def separable_taps(kernel):
    """Rank-1 factorisation kernel = outer(tv, th) of the table kernels (src/physics/kernels.py:13-28:
    exp(-(u^2+v^2)/2R^2)/sum and the uniform box are exactly separable). Returns (tv, th, residual)."""
    k = np.asarray(kernel, dtype=np.float64)
    tv, th = k.sum(1) / k.sum(), k.sum(0)   # rank 1: k[a,b] = rowsum[a] * colsum[b] / total
    return tv, th, float(np.abs(np.outer(tv, th) - k).max())


def blur_direct(x, kernel):
    """y[i,j] = sum_{a,b} k[a,b] x[(i-a+kh//2) mod H, (j-b+kw//2) mod W]."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    kh, kw = k.shape
    y = np.zeros_like(x)
    for a in range(kh):
        for b in range(kw):
            y += k[a, b] * np.roll(x, (a - kh // 2, b - kw // 2), axis=(-2, -1))
    return y


def blur_adjoint_direct(ct, kernel):
    """vjp of blur_direct = circular correlation: g[p,q] = sum k[a,b] ct[(p+a-kh//2) mod H, ...]."""
    ct = np.asarray(ct, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    kh, kw = k.shape
    g = np.zeros_like(ct)
    for a in range(kh):
        for b in range(kw):
            g += k[a, b] * np.roll(ct, (-(a - kh // 2), -(b - kw // 2)), axis=(-2, -1))
    return g


# ----------------------------------------------------------------------------------------------
# antialiased bicubic resize: Downsampling.A (src/physics/downsampling/__init__.py:16-19)
# = ATen _upsample_bicubic2d_aa; separable; per-axis dense weight matrix
# ----------------------------------------------------------------------------------------------
def _cubic(t, a):
    t = abs(t)
    if t <= 1.0:
        return ((a + 2.0) * t - (a + 3.0)) * t * t + 1.0
    if t < 2.0:
        return (((t - 5.0) * t + 8.0) * t - 4.0) * a
    return 0.0


def aa_out_size(n_in, rate):
    return int(floor(float(n_in) * (1.0 / rate)))


def aa_weights(n_in, rate):
    """(n_out, n_in) matrix of the antialiased bicubic (a=-0.5) filter for scale_factor = 1/rate."""
    n_out = aa_out_size(n_in, rate)
    scale = 1.0 / (1.0 / rate)
    support = 2.0 * scale if scale >= 1.0 else 2.0
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    Wm = np.zeros((n_out, n_in))
    for i in range(n_out):
        c = scale * (i + 0.5)
        lo = max(0, int(c - support + 0.5))
        hi = min(n_in, int(c + support + 0.5))
        w = np.array([_cubic((j - c + 0.5) * inv, -0.5) for j in range(lo, hi)])
        Wm[i, lo:hi] = w / w.sum()
    return Wm


def aa_downsample(x, rate):
    x = np.asarray(x, dtype=np.float64)
    Wv, Wh = aa_weights(x.shape[-2], rate), aa_weights(x.shape[-1], rate)
    return np.einsum("ai,...ij,bj->...ab", Wv, x, Wh)


def aa_downsample_adjoint(ct, rate, in_hw):
    ct = np.asarray(ct, dtype=np.float64)
    Wv, Wh = aa_weights(in_hw[0], rate), aa_weights(in_hw[1], rate)
    return np.einsum("ai,...ab,bj->...ij", Wv, ct, Wh)


def plain_bicubic_up_weights(n_in, rate):
    """F.interpolate(scale_factor=rate, mode='bicubic') (align_corners=False, a=-0.75), the
    deprecated A_adjoint at downsampling/__init__.py:33-34: (n_in*rate, n_in) matrix, clamped taps."""
    n_out = int(floor(n_in * float(rate)))
    scale = 1.0 / float(rate)
    Wm = np.zeros((n_out, n_in))
    for i in range(n_out):
        src = scale * (i + 0.5) - 0.5
        f = floor(src)
        t = src - f
        co = [_cubic(t + 1.0, -0.75), _cubic(t, -0.75), _cubic(1.0 - t, -0.75), _cubic(2.0 - t, -0.75)]
        for d in range(4):
            j = min(max(int(f) - 1 + d, 0), n_in - 1)
            Wm[i, j] += co[d]
    return Wm


# ----------------------------------------------------------------------------------------------
# scale transform: padded_downsampling_transform (src/transforms.py:60-83)
# = grid (src/transforms.py:27-43) + grid_sample(bicubic a=-0.75, reflection, align_corners=True)
# ----------------------------------------------------------------------------------------------
def _reflect_clip(v, n):
    if n == 1:
        return 0
    span = n - 1
    v = abs(v)
    flips = int(floor(v / span))
    extra = v - flips * span
    r = extra if flips % 2 == 0 else span - extra
    return int(min(max(r, 0), n - 1))


def scale_transform(x, rate, center):
    """x (B,C,H,W); rate (B,); center (B,2) = (cx, cy). Reproduces the `.view(1,h,w,2)` of the
    (w,h,2) stack for non-square inputs too (flat re-indexing)."""
    x = np.asarray(x, dtype=np.float64)
    B, C, H, W = x.shape
    out = np.zeros_like(x)
    A = -0.75
    for b in range(B):
        cx, cy, inv = float(center[b][0]), float(center[b][1]), 1.0 / float(rate[b])
        for i in range(H):
            for j in range(W):
                f = i * W + j
                aa, bb = f // H, f % H          # position in the (w,h) meshgrid
                gx = (2.0 / H * bb - 1.0 - cx) * inv + cx
                gy = (2.0 / W * aa - 1.0 - cy) * inv + cy
                ix = (gx + 1.0) / 2.0 * (W - 1)
                iy = (gy + 1.0) / 2.0 * (H - 1)
                fx, fy = floor(ix), floor(iy)
                tx, ty = ix - fx, iy - fy
                wx = [_cubic(tx + 1, A), _cubic(tx, A), _cubic(1 - tx, A), _cubic(2 - tx, A)]
                wy = [_cubic(ty + 1, A), _cubic(ty, A), _cubic(1 - ty, A), _cubic(2 - ty, A)]
                xs = [_reflect_clip(fx - 1 + d, W) for d in range(4)]
                ys = [_reflect_clip(fy - 1 + d, H) for d in range(4)]
                acc = np.zeros(C)
                for dy in range(4):
                    row = np.zeros(C)
                    for dx in range(4):
                        row += wx[dx] * x[b, :, ys[dy], xs[dx]]
                    acc += wy[dy] * row
                out[b, :, i, j] = acc
    return out


# ----------------------------------------------------------------------------------------------
# FFT "ideal" resamplers (src/models/convolutional.py:54-92, 113-133) as DFT-matrix products:
#   out = L1 @ X @ R1.T + L2 @ X @ R2.T   (separable rank 2, real)
# because each axis is an independent complex linear map (A on rows, B on the half-spectrum
# columns) and the final c2r takes the real part:  out = Re(A X B^T) = ReA X ReB^T - ImA X ImB^T.
# The reference discards its ifftshift (:89, :131), so the *shifted* spectrum is inverted as is;
# that quirk is reproduced here by construction.
# ----------------------------------------------------------------------------------------------
def _c2r_weights(N):
    w = np.full(N // 2 + 1, 2.0 / N)
    w[0] = 1.0 / N
    if N % 2 == 0:
        w[N // 2] = 1.0 / N
    return w


def ideal_down_axis(n, rate, half):
    """Complex (n, n) matrix of one axis of IdealDownsample *before* the ::rate subsampling.
    half=False: the full-FFT axis (dim -2); half=True: the rfft axis (dim -1)."""
    idx = np.arange(n)
    if not half:
        s, c = n // 2, ceil(n / (2 * rate))
        M = np.zeros((n, n), dtype=complex)
        for u in range(c, n - c):
            src = (u - s) % n
            M += np.outer(np.exp(2j * np.pi * idx * u / n), np.exp(-2j * np.pi * idx * src / n)) / n
        return M
    nh = n // 2 + 1
    s, c = nh // 2, ceil(nh / (2 * rate))
    w = _c2r_weights(n)
    M = np.zeros((n, n), dtype=complex)
    for v in range(c, nh - c):
        src = (v - s) % nh
        M += w[v] * np.outer(np.exp(2j * np.pi * idx * v / n), np.exp(-2j * np.pi * idx * src / n))
    return M


def ideal_up_axis(n, rate, half):
    """Complex (n*rate, n) matrix of one axis of IdealUpsample. Raises ValueError where the
    reference's slice assignment at convolutional.py:86 raises (shape mismatch)."""
    N = n * rate
    out_idx, in_idx = np.arange(N), np.arange(n)
    if not half:
        s = n // 2
        mv = (n * (rate - 1)) // 2
        top, bottom = mv + n % 2, mv
        if bottom == 0 or N - top - bottom != n:
            raise ValueError("IdealUpsample: spectrum does not fit its slot (reference raises too)")
        M = np.zeros((N, n), dtype=complex)
        for u in range(n):
            src = (u - s) % n
            M += np.outer(np.exp(2j * np.pi * out_idx * (u + top) / N),
                          np.exp(-2j * np.pi * in_idx * src / n)) / N
        return M
    nh = n // 2 + 1
    s = nh // 2
    mh = (nh * (rate - 1)) // 2
    left, right = mh + nh % 2, mh
    if right == 0 or nh * rate - left - right != nh:
        raise ValueError("IdealUpsample: spectrum does not fit its slot (reference raises too)")
    w = _c2r_weights(N)
    M = np.zeros((N, n), dtype=complex)
    for v in range(N // 2 + 1):
        if left <= v < left + nh:
            src = (v - left - s) % nh
            M += w[v] * np.outer(np.exp(2j * np.pi * out_idx * v / N),
                                 np.exp(-2j * np.pi * in_idx * src / n))
    return M


def ideal_down_matrices(H, W, rate=2):
    A = ideal_down_axis(H, rate, False)[::rate]
    Bm = ideal_down_axis(W, rate, True)[::rate]
    return A.real.copy(), Bm.real.copy(), -A.imag.copy(), Bm.imag.copy()


def ideal_up_matrices(H, W, rate=2):
    A = ideal_up_axis(H, rate, False)
    Bm = ideal_up_axis(W, rate, True)
    return A.real.copy(), Bm.real.copy(), -A.imag.copy(), Bm.imag.copy()


def sepmap2(x, L1, R1, L2, R2):
    x = np.asarray(x, dtype=np.float64)
    return (np.einsum("ai,...ij,bj->...ab", L1, x, R1) + np.einsum("ai,...ij,bj->...ab", L2, x, R2))


# ----------------------------------------------------------------------------------------------
# small pointwise pieces
# ----------------------------------------------------------------------------------------------
def gelu(x):
    from math import erf
    x = np.asarray(x, dtype=np.float64)
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


def layer_norm_channels(x, w, b, eps=1e-6):
    """x (B,C,H,W): normalise each pixel over C (biased variance), affine."""
    x = np.asarray(x, dtype=np.float64)
    mu = x.mean(1, keepdims=True)
    var = ((x - mu) ** 2).mean(1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * np.asarray(w).reshape(1, -1, 1, 1) + np.asarray(b).reshape(1, -1, 1, 1)
