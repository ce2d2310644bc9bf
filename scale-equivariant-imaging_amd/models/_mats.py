"""DFT-matrix form of the reference's FFT "ideal" resamplers (src/models/convolutional.py:54-92,
113-133), built once per (size, rate) on the host in float64.

The reference computes  irfft2( op( fftshift( rfft2(X) ) ) )  where `op` masks (down) or zero-embeds
(up) the *shifted* spectrum and the following `ifftshift` result is discarded (:89, :131). Each axis
is therefore an independent complex-linear map -- A on the full-FFT axis (rows), B on the half-spectrum
axis (columns) -- and the final complex-to-real transform takes a real part:

    out = Re(A X B^T) = Re(A) X Re(B)^T - Im(A) X Im(B)^T

a real, separable, rank-2 map, which `sei_sepmap2` applies as two small dense products per channel.
No FFT is evaluated anywhere in the product path.
"""
from functools import lru_cache
from math import ceil

import numpy as np
import torch


def _c2r_weights(n):
    w = np.full(n // 2 + 1, 2.0 / n)
    w[0] = 1.0 / n
    if n % 2 == 0:
        w[n // 2] = 1.0 / n
    return w


def _phase(rows, cols, n):
    return np.exp(2j * np.pi * np.outer(rows, cols) / n)


def _down_axis(n, rate, half):
    """One axis of IdealDownsample before the ::rate subsampling: complex (n, n)."""
    pos = np.arange(n)
    if not half:
        shift, cut = n // 2, ceil(n / (2 * rate))
        kept = np.arange(cut, n - cut)                 # bins of the shifted spectrum that survive
        src = (kept - shift) % n                       # the true frequency sitting in each of them
        return _phase(pos, kept, n) @ np.conj(_phase(pos, src, n)).T / n
    nh = n // 2 + 1
    shift, cut = nh // 2, ceil(nh / (2 * rate))
    kept = np.arange(cut, nh - cut)
    src = (kept - shift) % nh
    return (_phase(pos, kept, n) * _c2r_weights(n)[kept]) @ np.conj(_phase(pos, src, n)).T


def _up_axis(n, rate, half):
    """One axis of IdealUpsample: complex (n*rate, n). Raises where the reference's slice
    assignment (:86) raises."""
    big = n * rate
    out_pos, in_pos = np.arange(big), np.arange(n)
    if not half:
        shift = n // 2
        margin = (n * (rate - 1)) // 2
        first, last_margin = margin + n % 2, margin
        if last_margin == 0 or big - first - last_margin != n:
            raise RuntimeError("IdealUpsample: the spectrum does not fit its slot for this size and rate")
        slots = np.arange(n) + first
        src = (np.arange(n) - shift) % n
        return _phase(out_pos, slots, big) @ np.conj(_phase(in_pos, src, n)).T / big
    nh = n // 2 + 1
    shift = nh // 2
    margin = (nh * (rate - 1)) // 2
    first, last_margin = margin + nh % 2, margin
    if last_margin == 0 or nh * rate - first - last_margin != nh:
        raise RuntimeError("IdealUpsample: the spectrum does not fit its slot for this size and rate "
                           "(the reference raises here too: odd rate with a width divisible by 4)")
    slots = np.arange(nh) + first
    ok = slots <= big // 2                             # the c2r transform reads bins 0 .. big/2 only
    src = (np.arange(nh) - shift) % nh
    return (_phase(out_pos, slots[ok], big) * _c2r_weights(big)[slots[ok]]) @ np.conj(_phase(in_pos, src[ok], n)).T


@lru_cache(maxsize=None)
def _host_matrices(kind, H, W, rate):
    if kind == "down":
        A, Bm = _down_axis(H, rate, False)[::rate], _down_axis(W, rate, True)[::rate]
    else:
        A, Bm = _up_axis(H, rate, False), _up_axis(W, rate, True)
    return tuple(np.ascontiguousarray(m, dtype=np.float32) for m in (A.real, Bm.real, -A.imag, Bm.imag))


_DEVICE_CACHE = {}


def resample_matrices(kind, H, W, rate, device):
    """(L1, R1, L2, R2) and their transposes as float32 device tensors."""
    key = (kind, H, W, rate, str(device))
    if key not in _DEVICE_CACHE:
        host = _host_matrices(kind, H, W, rate)
        host_t = tuple(np.ascontiguousarray(m.T) for m in host)
        fwd = tuple(torch.from_numpy(m).to(device) for m in host) + pack_for_kernel(host, device)
        bwd = tuple(torch.from_numpy(m).to(device) for m in host_t) + pack_for_kernel(host_t, device)
        _DEVICE_CACHE[key] = (fwd, bwd)
    return _DEVICE_CACHE[key]


SM_PAD = 24          # padding of the output axis in the packed layouts (csrc/unet_kernels.hip)


def pack_for_kernel(mats, device):
    """(L1, R1, L2, R2), each (out, in) -> (RW, LH) for sei_sepmap2_packed:
    RW[j][j'][t] = R_t[j'][j] (j' padded to SM_PAD), LH[i][t][i'] = L_t[i'][i] (i' padded)."""
    L1, R1, L2, R2 = (np.asarray(m.detach().cpu() if isinstance(m, torch.Tensor) else m, dtype=np.float32)
                      for m in mats)
    Wo, Wi = R1.shape
    Ho, Hi = L1.shape
    wo_pad, ho_pad = -(-Wo // SM_PAD) * SM_PAD, -(-Ho // SM_PAD) * SM_PAD
    RW = np.zeros((Wi, wo_pad, 2), dtype=np.float32)
    RW[:, :Wo, 0], RW[:, :Wo, 1] = R1.T, R2.T
    LH = np.zeros((Hi, 2, ho_pad), dtype=np.float32)
    LH[:, 0, :Ho], LH[:, 1, :Ho] = L1.T, L2.T
    return torch.from_numpy(RW).to(device), torch.from_numpy(LH).to(device)


def constant_response(kind, H, W, rate, device, batch):
    """s[b, i', j'] = the map's output for an all-ones image, flattened to (batch * Ho * Wo,) float32.

    A bias added BEFORE the (linear, per-channel) resampler comes out as bias[c] * s[i', j']; it is not a
    constant because the reference's discarded ifftshift leaves the spectrum shifted (reference
    convolutional.py:131). Used to apply the 1x1 convolution after the ideal downsampler (_ops.DownsampleFn)."""
    key = ("ones", kind, H, W, rate, str(device), batch)
    if key not in _DEVICE_CACHE:
        L1, R1, L2, R2 = (m.astype(np.float64) for m in _host_matrices(kind, H, W, rate))
        resp = np.outer(L1.sum(1), R1.sum(1)) + np.outer(L2.sum(1), R2.sum(1))
        full = np.tile(resp.reshape(-1), batch).astype(np.float32)
        _DEVICE_CACHE[key] = torch.from_numpy(full).to(device)
    return _DEVICE_CACHE[key]
