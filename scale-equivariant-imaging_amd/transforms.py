"""Scale transforms of the equivariant-imaging loss (reference call surface: src/transforms.py).

The default ("padded") transform zooms each image out by a random rate about a random centre with
bicubic grid sampling and reflection padding. The reference materialises a (B,H,W,2) grid and calls
F.grid_sample; here `sei_scale_resample_fwd` generates the grid in-kernel from (rate, centre).
Random draws keep the reference's order and distributions (rand(B) then rand(B,2) on x's device).
"""
import math

import torch
from torch.nn import Module

import _native as N
from physics import _bands
from physics._ops import SeparableResampleOp, apply_linear


_TABLES = {}


def _table(values, device, dtype):
    """Device-resident copy of `values`, built once (a host->device copy is illegal inside hipGraph capture)."""
    key = (tuple(values), str(device), dtype)
    if key not in _TABLES:
        _TABLES[key] = torch.tensor(values, device=device, dtype=dtype)
    return _TABLES[key]


def sample_from(values, shape=(1,), dtype=torch.float32, device="cpu"):
    """Uniform draw from `values` (reference :5-12)."""
    table = _table(values, device, dtype)
    u = torch.rand(shape, device=device, dtype=dtype)
    return table[torch.floor(len(values) * u).to(torch.int)]


def sample_downsampling_parameters(image_count, device, dtype, rates, out=None):
    """Per-image (rate, centre in [-1,1]^2) (reference :15-24): rate = rates[floor(len(rates) u)], centre = 2 u' - 1 with
    u = rand(B) and u' = rand(B, 2) drawn in this order. On the GPU the two uniform draws are torch's (the generator is
    consumed exactly as by the reference's calls) and the arithmetic behind them -- scale, floor, table lookup, affine map:
    seven elementwise torch kernels -- is one launch (sei_scale_params; the same float32 operations, bit-identical).
    out = (rate (B,), centre (B, 1, 1, 2)): written in place (the static buffers of a captured step)."""
    if torch.device(device).type != "cuda" or dtype != torch.float32:
        rate = sample_from(rates, shape=(image_count,), dtype=dtype, device=device)
        center = (2 * torch.rand((image_count, 2), dtype=dtype, device=device) - 1).view(image_count, 1, 1, 2)
        if out is not None:
            out[0].copy_(rate)
            out[1].copy_(center)
            return out
        return rate, center
    table = _table(rates, device, dtype)
    u = torch.rand((image_count,), device=device, dtype=dtype)
    v = torch.rand((image_count, 2), device=device, dtype=dtype)
    rate, center = out if out is not None else (torch.empty_like(u), torch.empty((image_count, 1, 1, 2), device=device,
                                                                                   dtype=dtype))
    N.check_tensor(rate, "rate")
    N.check_tensor(center, "center")
    N.call("sei_scale_params", u.data_ptr(), v.data_ptr(), table.data_ptr(), len(rates), image_count, rate.data_ptr(),
           center.data_ptr())
    return rate, center


class _ScaleResample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, center, out_hw):
        N.check_tensor(x, "x")
        B, C, Hi, Wi = x.shape
        H, W = out_hw
        rate = N.check_tensor(rate.reshape(B).contiguous(), "downsampling_rate")
        center = N.check_tensor(center.reshape(B, 2).contiguous(), "center")
        y = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
        N.call("sei_scale_resample_fwd", x.data_ptr(), y.data_ptr(), rate.data_ptr(), center.data_ptr(),
               B, C, Hi, Wi, H, W)
        ctx.save_for_backward(rate, center)
        ctx.in_hw = (Hi, Wi)
        return y

    @staticmethod
    def backward(ctx, g):
        rate, center = ctx.saved_tensors
        g = g.contiguous()
        B, C, H, W = g.shape
        Hi, Wi = ctx.in_hw
        gx = torch.zeros((B, C, Hi, Wi), dtype=g.dtype, device=g.device)
        N.call("sei_scale_resample_bwd", g.data_ptr(), gx.data_ptr(), rate.data_ptr(), center.data_ptr(),
               B, C, Hi, Wi, H, W)
        return gx, None, None, None


_AA_OPS = {}


def _aa_resize(x, scale_factor, antialias=True):
    """F.interpolate(x, scale_factor, 'bicubic', antialias) as a banded separable product."""
    key = (float(scale_factor), bool(antialias))
    if key not in _AA_OPS:
        build = _bands.aa_bicubic_matrix if antialias else _bands.plain_bicubic_matrix
        _AA_OPS[key] = SeparableResampleOp(lambda n, s=key[0], f=build: f(n, s))
    return apply_linear(_AA_OPS[key], x.contiguous())


def alias_free_interpolate(x, downsampling_rate, interpolation_mode):
    """Reference :46-57: a per-image antialiased resize, then torch.stack -- which, as in the
    reference, only succeeds when every image of the batch drew the same rate."""
    if interpolation_mode != "bicubic":
        raise ValueError("only mode='bicubic' is supported")
    rates = [float(r) for r in downsampling_rate.reshape(-1).tolist()]    # host sync, as .item() does
    return torch.stack([_aa_resize(x[i:i + 1], rates[i])[0] for i in range(x.shape[0])])


def padded_downsampling_transform(x, downsampling_rate, center, mode, padding_mode, antialiased):
    """Reference :60-83. With `antialiased` the pre-filtered (smaller) image is sampled on the grid
    of the ORIGINAL shape, as the reference does (the output keeps the original size)."""
    if mode != "bicubic" or padding_mode != "reflection":
        raise ValueError("only mode='bicubic', padding_mode='reflection' are supported")
    out_hw = tuple(x.shape[-2:])
    if antialiased:
        x = alias_free_interpolate(x, downsampling_rate=downsampling_rate, interpolation_mode=mode)
    return _ScaleResample.apply(x.contiguous(), downsampling_rate, center, out_hw)


class PaddedDownsamplingTransform(Module):
    def __init__(self, antialias, downsampling_rates):
        super().__init__()
        self.antialias = antialias
        self.downsampling_rates = downsampling_rates

    def sample(self, image_count, device, dtype):
        return sample_downsampling_parameters(image_count=image_count, device=device, dtype=dtype,
                                              rates=self.downsampling_rates)

    def sample_into(self, rate, center):
        sample_downsampling_parameters(image_count=rate.numel(), device=rate.device, dtype=rate.dtype,
                                       rates=self.downsampling_rates, out=(rate, center))

    def forward(self, x, params=None):
        """params: (rate (B,), centre (B,1,1,2)) drawn by the caller with `sample`; drawn here otherwise."""
        rate, center = params if params is not None else self.sample(x.shape[0], x.device, x.dtype)
        return padded_downsampling_transform(x, downsampling_rate=rate, center=center,
                                             antialiased=self.antialias, mode="bicubic",
                                             padding_mode="reflection")


def normal_downsampling_transform(x, downsampling_rate, mode, antialiased):
    """Reference :112-123: every image resized by the same python-float rate."""
    if mode != "bicubic":
        raise ValueError("only mode='bicubic' is supported")
    return _aa_resize(x, downsampling_rate, antialiased)


class NormalDownsamplingTransform(Module):
    def __init__(self, antialias, downsampling_rates):
        super().__init__()
        self.antialias = antialias
        self.downsampling_rates = downsampling_rates

    def forward(self, x):
        rate = sample_from(self.downsampling_rates, shape=(), dtype=x.dtype, device=x.device).item()
        return normal_downsampling_transform(x, downsampling_rate=rate, mode="bicubic",
                                             antialiased=self.antialias)


class ScalingTransform(Module):
    def __init__(self, kind, antialias):
        super().__init__()
        rates = [0.75, 0.5]
        kinds = {"padded": PaddedDownsamplingTransform, "normal": NormalDownsamplingTransform}
        if kind not in kinds:
            raise ValueError(f"Unknown kind: {kind}")
        self.kind, self.antialias = kind, antialias
        self.transform = kinds[kind](antialias=antialias, downsampling_rates=rates)

    def sample(self, image_count, device, dtype):
        """The padded kind's per-image draws (rate, centre), in the reference's order (:15-24)."""
        return self.transform.sample(image_count, device, dtype)

    def sample_into(self, rate, center):
        """`sample` written into existing buffers; only the padded kind draws on the device."""
        self.transform.sample_into(rate, center)

    def forward(self, x, params=None):
        return self.transform(x) if params is None else self.transform(x, params=params)


class Shift(Module):
    """Random circular shift (deepinv.transform.Shift at v0.2.0, the `Shifts` option of
    src/losses/__init__.py:92-94; restated from its documented behaviour -- deepinv is absent, unpinned): one
    shift per axis, drawn without replacement from [-shift_max*size, shift_max*size) by a random permutation
    (CPU generator: randperm over H first, then W), applied with torch.roll to the whole batch."""

    def __init__(self, n_trans=1, shift_max=1.0):
        super().__init__()
        self.n_trans, self.shift_max = n_trans, shift_max

    def forward(self, x):
        H, W = x.shape[-2:]
        assert self.n_trans <= H - 1 and self.n_trans <= W - 1
        h_max, w_max = int(self.shift_max * H), int(self.shift_max * W)
        sx = torch.arange(-h_max, h_max)[torch.randperm(2 * h_max)][: self.n_trans]
        sy = torch.arange(-w_max, w_max)[torch.randperm(2 * w_max)][: self.n_trans]
        return torch.cat([torch.roll(x, [int(a), int(b)], [-2, -1]) for a, b in zip(sx, sy)], dim=0)


class _RotateNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, theta):
        N.check_tensor(x, "x")
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        N.call("sei_rotate_nearest_fwd", x.data_ptr(), y.data_ptr(), B * C, H, W, *theta)
        ctx.theta = theta
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, C, H, W = g.shape
        gx = torch.zeros_like(g)
        N.call("sei_rotate_nearest_bwd", g.data_ptr(), gx.data_ptr(), B * C, H, W, *ctx.theta)
        return gx, None


class Rotate(Module):
    """Random rotation about the image centre (deepinv.transform.Rotate at v0.2.0, the `Rotations` option of
    src/losses/__init__.py:84-91; restated from recollection of its source -- deepinv and torchvision are absent,
    unpinned): with the default group (degrees=360) one whole-degree angle per transform is taken from 1..359 by a
    random permutation on the CPU generator, and the batch is rotated by torchvision's `rotate` defaults (nearest
    neighbour, same canvas, zero fill), here `sei_rotate_nearest_fwd`."""

    def __init__(self, n_trans=1, degrees=360):
        super().__init__()
        self.n_trans, self.group_size = n_trans, degrees

    def sample(self):
        if self.group_size == 360:
            angles = torch.arange(0, 360)[1:][torch.randperm(359)]
        else:
            angles = torch.arange(0, 360, int(360 / (self.group_size + 1)))[1:][torch.randperm(self.group_size)]
        return [float(a) for a in angles[: self.n_trans]]

    @staticmethod
    def matrix(angle):
        """Linear part of torchvision's inverse affine matrix for rotate(img, angle), as float32 values."""
        a = math.radians(angle)
        return tuple(float(torch.tensor(v, dtype=torch.float32)) for v in
                     (math.cos(a), -math.sin(a), math.sin(a), math.cos(a)))

    def forward(self, x, params=None):
        angles = self.sample() if params is None else params
        x = x.contiguous()
        return torch.cat([_RotateNearest.apply(x, self.matrix(a)) for a in angles], dim=0)


class CombinedTransform(Module):
    def __init__(self, transforms):
        super().__init__()
        self.transforms = transforms

    def forward(self, x):
        for t in self.transforms:
            x = t(x)
        return x
