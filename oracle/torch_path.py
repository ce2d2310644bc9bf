"""ORACLE (test infrastructure) -- torch-CPU restatement of the reference hot path.

Every function runs the same torch op sequence as the reference function it cites (paths under
/root/reference), written functionally (tensors and a flat state_dict in, tensors out) so that it
can be driven with injected randomness. Works in float32 or float64 on CPU. This is what
`bench.py` times as `cpu_baseline` (kind "port") and what the GPU parity tests compare against.
"""
from math import ceil

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# physics
# ----------------------------------------------------------------------------------------------
KERNEL_TABLE = {  # src/physics/kernels.py:3-10
    "Gaussian_R1": ("gaussian", 1), "Gaussian_R2": ("gaussian", 2), "Gaussian_R3": ("gaussian", 3),
    "Box_R2": ("box", 2), "Box_R3": ("box", 3), "Box_R4": ("box", 4),
}


def blur_kernel(name):
    """src/physics/kernels.py:13-28 -- float64 (k,k) kernel, sum 1."""
    assert name in KERNEL_TABLE, f"Unsupported kernel: {name}"
    family, level = KERNEL_TABLE[name]
    if family == "box":
        n = 2 * level + 1
        return torch.full((n, n), 1.0 / (n * n), dtype=torch.float64) * 1.0
    n = 6 * level + 1
    t = torch.arange(n, dtype=torch.float64) - (n - 1) / 2
    g = torch.exp(-(t[:, None] ** 2 + t[None, :] ** 2) / (2 * level**2))
    return g / g.sum()


def blur_fft(x, kernel):
    """src/physics/blur/__init__.py:205-223 (BlurV2.A): zero-padded PSF rolled to the origin,
    rfft2 * rfft2, irfft2. `kernel` is (k,k) or (1,1,k,k), any float dtype (cast to x's)."""
    H, W = x.shape[-2:]
    k = kernel.reshape(kernel.shape[-2], kernel.shape[-1]).to(x.dtype)
    psf = x.new_zeros((H, W))
    psf[: k.shape[0], : k.shape[1]] = k
    psf = torch.roll(psf, shifts=(-(k.shape[0] // 2), -(k.shape[1] // 2)), dims=(0, 1))
    spec = torch.fft.rfft2(x) * torch.fft.rfft2(psf)
    return torch.fft.irfft2(spec, s=(H, W))


def downsample_aa(x, rate):
    """src/physics/downsampling/__init__.py:16-19 (Downsampling.A)."""
    return F.interpolate(x, scale_factor=1 / rate, mode="bicubic", antialias=True)


def upsample_plain_bicubic(y, rate):
    """src/physics/downsampling/__init__.py:33-34 (deprecated A_adjoint)."""
    return F.interpolate(y, scale_factor=rate, mode="bicubic")


def vjp_of(fn, in_shape, dtype=torch.float32):
    """deepinv `adjoint_function` as used at blur/__init__.py:226, downsampling/__init__.py:30."""
    probe = torch.ones(in_shape, dtype=dtype)
    _, pull = torch.func.vjp(fn, probe)
    return lambda ct: pull(ct)[0]


def add_noise(y, sigma, n=None):
    """deepinv GaussianNoise [recollection, unpinned]: y + sigma * N(0,1)."""
    if n is None:
        n = torch.randn_like(y)
    return y + sigma * n


# ----------------------------------------------------------------------------------------------
# crop (src/crop.py:8-57), with the 4-D batch quirk of MinSizePadding
# ----------------------------------------------------------------------------------------------
def _min_size_pad(t, size):
    # src/crop.py:49-57: reads shape[1], shape[2] -- C and H for a 4-D batch, H and W for a 3-D item
    pad_bottom = max(0, size - t.shape[1])
    pad_right = max(0, size - t.shape[2])
    return F.pad(t, (0, pad_right, 0, pad_bottom))


def crop_pair_bounds(x_shape, y_shape, size, ratio):
    """Shapes after padding and the exclusive upper bounds of the two randint draws."""
    def padded(shape, s):
        hp = max(0, s - shape[1])
        wp = max(0, s - shape[2])
        return tuple(shape[:-2]) + (shape[-2] + hp, shape[-1] + wp)
    xs, ys = padded(x_shape, size * ratio), padded(y_shape, size)
    return xs, ys, ys[-2] - size + 1, ys[-1] - size + 1


def crop_pair(x, y, size, ratio, i=None, j=None):
    """src/crop.py:15-39 with location='random'; i, j injectable (else two torch.randint draws)."""
    x = _min_size_pad(x, size * ratio)
    y = _min_size_pad(y, size)
    h, w = y.shape[-2:]
    if i is None:
        i = torch.randint(0, h - size + 1, size=(1,)).item()
        j = torch.randint(0, w - size + 1, size=(1,)).item()
    xs = x[..., i * ratio: i * ratio + size * ratio, j * ratio: j * ratio + size * ratio]
    ys = y[..., i: i + size, j: j + size]
    return xs, ys


# ----------------------------------------------------------------------------------------------
# scale transform (src/transforms.py)
# ----------------------------------------------------------------------------------------------
def sample_scale_params(count, device="cpu", dtype=torch.float32, rates=(0.75, 0.5)):
    """src/transforms.py:5-24 -- draw order: rand(count) then rand(count, 2)."""
    vals = torch.tensor(list(rates), device=device, dtype=dtype)
    u = torch.rand((count,), device=device, dtype=dtype)
    idx = torch.floor(len(rates) * u).to(torch.int)
    center = 2 * torch.rand((count, 2), device=device, dtype=dtype) - 1
    return vals[idx], center.view(count, 1, 1, 2)


def scale_grid(shape, rate, center, dtype):
    """src/transforms.py:27-43 -- note: 2/w*arange(w)-1 (not the align-corners grid), 'ij'
    meshgrid of (u over w, v over h) then stack([V, U]) viewed as (1,h,w,2)."""
    b, _, h, w = shape
    u = 2 / w * torch.arange(w, dtype=dtype) - 1
    v = 2 / h * torch.arange(h, dtype=dtype) - 1
    uu, vv = torch.meshgrid(u, v, indexing="ij")
    g = torch.stack([vv, uu], dim=-1).view(1, h, w, 2).repeat(b, 1, 1, 1)
    c = center.view(b, 1, 1, 2)
    return 1 / rate.view(b, 1, 1, 1) * (g - c) + c


def scale_transform(x, rate, center, antialias=False):
    """src/transforms.py:60-83 (padded kind)."""
    shape = x.shape
    if antialias:  # src/transforms.py:46-57
        x = torch.stack([
            F.interpolate(x[k:k + 1], scale_factor=rate[k].item(), mode="bicubic", antialias=True)[0]
            for k in range(x.shape[0])])
    g = scale_grid(shape, rate, center, x.dtype)
    return F.grid_sample(x, g, mode="bicubic", padding_mode="reflection", align_corners=True)


def scale_transform_normal(x, rate, antialias):
    """src/transforms.py:112-123 (normal kind, one python-float rate for the whole batch)."""
    return torch.stack([
        F.interpolate(x[k:k + 1], scale_factor=rate, mode="bicubic", antialias=antialias)[0]
        for k in range(x.shape[0])])


def rotate_nearest(x, angle):
    """deepinv.transform.Rotate's resampler (src/losses/__init__.py:84-91 -> torchvision rotate(x, angle) with its
    defaults) [torchvision absent -> restated from recollection of its tensor path, UNPINNED]: inverse affine matrix
    (cos a, -sin a, 0; sin a, cos a, 0) in float32, base grid of pixel centres about the image centre, rows divided
    by (W/2, H/2), one bmm, then F.grid_sample(nearest, zeros, align_corners=False)."""
    import math
    B, C, H, W = x.shape
    a = math.radians(angle)
    theta = torch.tensor([math.cos(a), -math.sin(a), 0.0, math.sin(a), math.cos(a), 0.0], dtype=x.dtype).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3, dtype=x.dtype)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H).unsqueeze(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=x.dtype)
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2).expand(B, H, W, 2)
    return F.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)


def rotate_source_margin(shape, angle):
    """Distance of every output pixel's source coordinate from the nearest rounding tie (.5), in float64: where this is
    tiny, float32 evaluation order decides the neighbour and two correct implementations may differ."""
    import math
    H, W = shape
    a = math.radians(angle)
    j = torch.arange(W, dtype=torch.float64) + 0.5 - W / 2
    i = (torch.arange(H, dtype=torch.float64) + 0.5 - H / 2).unsqueeze(-1)
    ix = math.cos(a) * j - math.sin(a) * i + W / 2 - 0.5
    iy = math.sin(a) * j + math.cos(a) * i + H / 2 - 0.5
    tie = lambda v: ((v - torch.floor(v)) - 0.5).abs()
    return torch.minimum(tie(ix), tie(iy))


# ----------------------------------------------------------------------------------------------
# U-Net (src/models/convolutional.py), functional over the reference's state_dict key layout
# ----------------------------------------------------------------------------------------------
def _ln_channels(x, w, b):
    # convolutional.py:21-30: LayerNorm(C, eps=1e-6) over the channel axis of NCHW
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, 1e-6).permute(0, 3, 1, 2)


def ideal_downsample(x, rate=2):
    """convolutional.py:113-133; the ifftshift at :131 is a discarded no-op and is omitted."""
    H, W = x.shape[-2:]
    f = torch.fft.fftshift(torch.fft.rfft2(x), dim=(-2, -1))
    ch, cw = ceil(f.shape[-2] / (2 * rate)), ceil(f.shape[-1] / (2 * rate))
    keep = torch.zeros_like(f)
    keep[..., ch:-ch, cw:-cw] = 1
    return torch.fft.irfft2(keep * f, s=(H, W))[..., ::rate, ::rate]


def ideal_upsample(x, rate=2):
    """convolutional.py:54-92; the ifftshift at :89 is a discarded no-op and is omitted."""
    H, W = x.shape[-2:]
    f = torch.fft.fftshift(torch.fft.rfft2(x), dim=(-2, -1))
    fh, fw = f.shape[-2:]
    big = f.new_zeros(f.shape[:-2] + (fh * rate, fw * rate))
    mv, mh = (fh * (rate - 1)) // 2, (fw * (rate - 1)) // 2
    top, bottom = mv + (fh % 2), mv
    left, right = mh + (fw % 2), mh
    big[..., top:-bottom, left:-right] = f
    return torch.fft.irfft2(big, s=(H * rate, W * rate))


def _conv_block(sd, p, x):
    # convolutional.py:44-51
    C = x.shape[1]
    h = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=3, groups=C)
    h = _ln_channels(h, sd[p + "ln.ln.weight"], sd[p + "ln.ln.bias"])
    h = F.gelu(F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"]))
    return x + F.conv2d(h, sd[p + "conv3.weight"], sd[p + "conv3.bias"])


def _upsample_layer(sd, p, x, rate):
    # convolutional.py:95-110: seq = [IdealUpsample, LayerNorm, Conv2d 1x1]
    x = ideal_upsample(x, rate)
    x = _ln_channels(x, sd[p + "seq.1.ln.weight"], sd[p + "seq.1.ln.bias"])
    return F.conv2d(x, sd[p + "seq.2.weight"], sd[p + "seq.2.bias"])


def _downsample_layer(sd, p, x):
    # convolutional.py:146-150
    x = _ln_channels(x, sd[p + "ln.ln.weight"], sd[p + "ln.ln.bias"])
    return ideal_downsample(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"]), 2)


def unet_forward(sd, y, *, scales, upsampling_rate=1, residual=True, inner_residual=True,
                 num_conv_blocks=1, inout_convs=True, **_unused):
    """ConvolutionalModel.forward (convolutional.py:286-303) + UNet.forward (:214-249)."""
    div = 2 ** (scales - 1)
    ph, pw = (-y.shape[-2]) % div, (-y.shape[-1]) % div
    if ph or pw:
        y = F.pad(y, (0, pw, 0, ph), mode="reflect")
    x = y
    net = "seq.0."
    if upsampling_rate != 1:
        x = _upsample_layer(sd, "seq.0.", x, upsampling_rate)
        net = "seq.1."
    x0 = x
    if inout_convs:
        x = F.conv2d(x, sd[net + "in_conv.weight"], sd[net + "in_conv.bias"], padding=1)
    skips, seq = [], 0

    def blocks(t, s):
        for k in range(num_conv_blocks):
            t = _conv_block(sd, f"{net}conv_sequences.{s}.{k}.", t)
        return t

    for lvl in range(scales - 1):
        xb = x
        x = blocks(x, seq)
        seq += 1
        if inner_residual:
            x = x + xb
        skips.append(x)
        x = _downsample_layer(sd, f"{net}downsampling_layers.{lvl}.", x)
    x = blocks(x, seq)
    seq += 1
    for lvl in range(scales - 1):
        x = _upsample_layer(sd, f"{net}upsampling_layers.{lvl}.", x, 2)
        x = x + skips.pop()
        x = blocks(x, seq)
        seq += 1
    if inout_convs:
        x = F.conv2d(x, sd[net + "out_conv.weight"], sd[net + "out_conv.bias"], padding=1)
    if residual:
        x = x + x0
    # convolutional.py:296-301
    if ph and pw:
        x = x[:, :, :-ph, :-pw]
    elif ph:
        x = x[:, :, :-ph, :]
    elif pw:
        x = x[:, :, :, :-pw]
    return x


def unet_init_state_dict(hidden_channels=32, scales=5, upsampling_rate=1, num_conv_blocks=1,
                         in_channels=3, seed=0, dtype=torch.float32):
    """A state_dict with the reference's key layout and torch's default Conv2d/LayerNorm init
    distributions (kaiming_uniform(a=sqrt(5)) + uniform bias; ones/zeros), seeded. The draw ORDER
    differs from nn.Module construction, so this is for synthetic benchmarks and tests that share
    the dict with the product -- not a reproduction of `torch.manual_seed(0); ConvolutionalModel()`."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin_per_group, k):
        fan_in = cin_per_group * k * k
        bound = 1.0 / fan_in**0.5
        sd[name + ".weight"] = ((torch.rand((cout, cin_per_group, k, k), generator=g) * 2 - 1) * bound).to(dtype)
        sd[name + ".bias"] = ((torch.rand((cout,), generator=g) * 2 - 1) * bound).to(dtype)

    def ln(name, c):
        sd[name + ".weight"] = torch.ones(c, dtype=dtype)
        sd[name + ".bias"] = torch.zeros(c, dtype=dtype)

    net = "seq.0."
    if upsampling_rate != 1:
        ln("seq.0.seq.1.ln", in_channels)
        conv("seq.0.seq.2", in_channels, in_channels, 1)
        net = "seq.1."
    conv(net + "in_conv", hidden_channels, in_channels, 3)
    conv(net + "out_conv", in_channels, hidden_channels, 3)

    def block(prefix, c):
        conv(prefix + "conv1", c, 1, 7)
        ln(prefix + "ln.ln", c)
        conv(prefix + "conv2", 4 * c, c, 1)
        conv(prefix + "conv3", c, 4 * c, 1)

    c, seq = hidden_channels, 0
    for lvl in range(scales - 1):
        for k in range(num_conv_blocks):
            block(f"{net}conv_sequences.{seq}.{k}.", c)
        seq += 1
        ln(f"{net}downsampling_layers.{lvl}.ln.ln", c)
        conv(f"{net}downsampling_layers.{lvl}.conv", 4 * c, c, 1)
        c *= 4
    for k in range(num_conv_blocks):
        block(f"{net}conv_sequences.{seq}.{k}.", c)
    seq += 1
    for lvl in range(scales - 1):
        ln(f"{net}upsampling_layers.{lvl}.seq.1.ln", c)
        conv(f"{net}upsampling_layers.{lvl}.seq.2", c // 4, c, 1)
        c //= 4
        for k in range(num_conv_blocks):
            block(f"{net}conv_sequences.{seq}.{k}.", c)
        seq += 1
    return sd


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def sure_loss(y, x_net, A, model, sigma, *, margin=0, tau=1e-2, cropped_div=True,
              averaged_cst=None, b=None):
    """src/losses/sure.py:48-76 + mc_div :7-32. `b` = the injected N(0,1) draw on the interior
    (shape (B,C,H-2m,W-2m)); drawn with torch.randn if None."""
    m = margin if cropped_div else 0
    y1 = A(x_net)
    if b is None:
        b = torch.randn((y.shape[0], y.shape[1], y.shape[2] - 2 * m, y.shape[3] - 2 * m), dtype=y.dtype)
    if m:
        bb = torch.zeros_like(y)
        bb[:, :, m:-m, m:-m] = b
    else:
        bb = b
    y2 = A(model(y + bb * tau))
    d = bb * (y2 - y1) / tau
    if m:
        d = d[:, :, m:-m, m:-m]
    div = 2 * sigma**2 * d.mean()
    r = y1 - y
    if margin:
        r = r[:, :, margin:-margin, margin:-margin]
    mse = r.pow(2).mean()
    cst = sigma**2 if averaged_cst else sigma**2 / y.shape[0]
    return mse + div - cst


def ei_loss(x_net, A, model, transform, sigma, *, alpha=1.0, stop_gradient=True, n=None):
    """deepinv v0.2.0 EILoss as configured at src/losses/__init__.py:117-122 [recollection,
    unpinned]: x2 = T(x_net) (under no_grad), y2 = A(x2) + sigma*n, x3 = model(y2),
    alpha * mean((x3 - x2)^2). Returns (loss, x2, x3)."""
    if stop_gradient:
        with torch.no_grad():
            x2 = transform(x_net)
    else:
        x2 = transform(x_net)
    y2 = add_noise(A(x2), sigma, n)
    x3 = model(y2)
    return alpha * F.mse_loss(x3, x2), x2, x3


def r2r_ei_loss(y, A, model, transform, sigma, *, unit_pert, n1, n2, stop_gradient=True):
    """src/losses/r2r.py (in-tree): R2RLoss(eta=sigma, alpha=0.5) + the EI term with consistent input noise
    (:9-57), with the three normal draws injected. Returns (loss, r2r_term, ei_term)."""
    alpha = 0.5
    pert = unit_pert * sigma
    out = model(y + pert * alpha)
    l_r2r = F.mse_loss(A(out), y - pert / alpha)
    x1 = model(y + 0.5 * sigma * n1)
    if stop_gradient:
        with torch.no_grad():
            x2 = transform(x1)
    else:
        x2 = transform(x1)
    x3 = model(A(x2) + 1.5 * sigma * n2)
    l_ei = F.mse_loss(x3, x2)
    return l_r2r + l_ei, l_r2r, l_ei


def proposed_loss(y, A, model, sigma, *, margin, rate, center, b=None, n=None, alpha=1.0,
                  stop_gradient=True, averaged_cst=None, cropped_div=True):
    """src/losses/__init__.py:133-142 (ProposedLoss.forward) with the default loss list
    [SureGaussianLoss, EILoss] and the padded scale transform; all randomness injectable."""
    x_net = model(y)
    l_sure = sure_loss(y, x_net, A, model, sigma, margin=margin, cropped_div=cropped_div,
                       averaged_cst=averaged_cst, b=b)
    l_ei, x2, x3 = ei_loss(x_net, A, model, lambda t: scale_transform(t, rate, center), sigma,
                           alpha=alpha, stop_gradient=stop_gradient, n=n)
    return l_sure + l_ei, dict(x_net=x_net, x2=x2, x3=x3, loss_sure=l_sure, loss_ei=l_ei)


# ----------------------------------------------------------------------------------------------
# schedule + metric
# ----------------------------------------------------------------------------------------------
def lr_factor(epoch, epochs, kind="delayed_linear_decay"):
    """Closed form of src/scheduler.py:5-22 as torch's SequentialLR/LinearLR/MultiStepLR evaluate
    it (pinned by golden G9): multiplier on the base LR in effect during 0-based `epoch`."""
    if kind == "multi_step_decay":
        ms = [epochs * 50 // 100, epochs * 80 // 100, epochs * 90 // 100, epochs * 95 // 100]
        return 0.5 ** sum(1 for m in ms if epoch >= m)
    half = epochs // 2
    if epoch < half:
        return 1.0
    span = half - 1
    t = min(epoch - half, span)
    return 1.0 + (1e-2 - 1.0) * t / span if span > 0 else 1e-2


def psnr_y(x_hat, x):
    """src/metrics.py:10-13 [kornia rgb_to_ycbcr + torchmetrics PSNR, unpinned]: Y = .299R+.587G+.114B,
    10*log10(1 / mse) over a single (3,H,W) image."""
    wts = torch.tensor([0.299, 0.587, 0.114], dtype=x.dtype).view(3, 1, 1)
    d = ((x_hat - x) * wts).sum(0)
    return 10 * torch.log10(1.0 / d.pow(2).mean())
