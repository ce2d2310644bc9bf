"""ConvNeXt-style U-Net (reference call surface: src/models/convolutional.py).

Same module tree, parameter names, shapes and construction order as the reference, so that
`state_dict()` / `load_state_dict()` interchange `weights.pt` files with it and
`torch.manual_seed(0); ConvolutionalModel(...)` draws the same initial weights. The torch layer
classes (Conv2d, LayerNorm) are used as PARAMETER CONTAINERS only -- their forward is never called;
every forward/backward runs in libsei_hip.so on NHWC activations (models/_ops.py).

After construction or a device move the parameters are re-homed into one contiguous float32 bucket
(`flat_params`) with a matching gradient bucket (`flat_grads`): one fused Adam launch and one RCCL
all-reduce cover the whole model.
"""
import torch
import torch.nn.functional as F
from torch.nn import GELU, Conv2d, LayerNorm as BaseLayerNorm, Module, ModuleList, Sequential

import _native as N
from . import _mats, _ops


class LayerNorm(Module):
    """LayerNorm over the channels of each pixel (reference :21-30); callable on NHWC activations."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.ln = BaseLayerNorm(*args, **kwargs)

    def forward(self, x):
        B, H, W, C = x.shape
        return _LayerNormFn.apply(x, self.ln.weight, self.ln.bias)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        B, H, W, C = x.shape
        y, mean, rstd = _ops.layer_norm(N.check_tensor(x, "x").view(-1, C), gamma, beta)
        ctx.save_for_backward(x, mean, rstd)
        ctx.params = (gamma, beta)
        return y.view(B, H, W, C)

    @staticmethod
    def backward(ctx, go):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        C = x.shape[-1]
        gx = _ops.layer_norm_bwd(x.view(-1, C), gamma, mean, rstd, go.contiguous().view(-1, C),
                                 _ops.grad_of(gamma), _ops.grad_of(beta))
        return gx.view_as(x), None, None


class ConvBlock(Module):
    def __init__(self, dim):
        super().__init__()
        self.conv1 = Conv2d(in_channels=dim, out_channels=dim, kernel_size=7, padding=3, groups=dim)
        self.ln = LayerNorm(dim, eps=1e-6)
        self.conv2 = Conv2d(in_channels=dim, out_channels=4 * dim, kernel_size=1)
        self.gelu = GELU()
        self.conv3 = Conv2d(in_channels=4 * dim, out_channels=dim, kernel_size=1)

    def forward(self, x, twice=False):
        fn = _ops.ConvBlockFn16 if _ops.use_bf16_blocks(x.shape[-1]) else _ops.ConvBlockFn
        return fn.apply(x, self.conv1.weight, self.conv1.bias, self.ln.ln.weight, self.ln.ln.bias,
                                      self.conv2.weight, self.conv2.bias, self.conv3.weight, self.conv3.bias,
                                      twice)


class _SepMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, rate):
        B, H, W, C = N.check_tensor(x, "x").shape
        fwd, bwd = _mats.resample_matrices(kind, H, W, rate, x.device)
        ctx.mats_t, ctx.hw = bwd, (H, W)
        return _ops.sepmap2(x, fwd, fwd[0].shape[0], fwd[1].shape[0])

    @staticmethod
    def backward(ctx, go):
        return _ops.sepmap2(go.contiguous(), ctx.mats_t, *ctx.hw), None, None


class IdealUpsample(Module):
    def __init__(self, rate=2):
        super().__init__()
        self.rate = rate

    def forward(self, x):
        return _SepMapFn.apply(x, "up", self.rate)


class IdealDownsample(Module):
    def __init__(self, rate=2):
        super().__init__()
        self.rate = rate

    def forward(self, x):
        return _SepMapFn.apply(x, "down", self.rate)


class Upsample(Module):
    def __init__(self, in_channels, out_channels=None, rate=2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels or in_channels // (rate**2)
        self.rate = rate
        self.seq = Sequential()
        self.seq.append(IdealUpsample(rate=self.rate))
        self.seq.append(LayerNorm(self.in_channels, eps=1e-6))
        self.seq.append(Conv2d(self.in_channels, self.out_channels, kernel_size=1, stride=1))

    def forward(self, x, skip=None):
        ln, conv = self.seq[1].ln, self.seq[2]
        fn = _ops.UpsampleFn16 if _ops.use_bf16_blocks(x.shape[-1]) else _ops.UpsampleFn
        return fn.apply(x, skip, ln.weight, ln.bias, conv.weight, conv.bias, self.rate)


class Downsample(Module):
    def __init__(self, in_channels, out_channels=None, rate=2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels or in_channels * (rate**2)
        self.rate = rate
        self.ln = LayerNorm(self.in_channels, eps=1e-6)
        self.conv = Conv2d(self.in_channels, self.out_channels, kernel_size=1, stride=1)
        self.ideal_downsample = IdealDownsample(rate=self.rate)

    def forward(self, x):
        fn = _ops.DownsampleFn16 if _ops.use_bf16_blocks(x.shape[-1]) else _ops.DownsampleFn
        return fn.apply(x, self.ln.ln.weight, self.ln.ln.bias, self.conv.weight, self.conv.bias, self.rate)


class UNet(Module):
    def __init__(self, in_channels, hidden_channels, inout_convs, scales, num_conv_blocks, rate, residual,
                 inner_residual):
        super().__init__()
        self.scales = scales
        self.residual = residual
        self.inner_residual = inner_residual
        self.conv_sequences = ModuleList()
        self.downsampling_layers = ModuleList()
        self.upsampling_layers = ModuleList()

        width = in_channels
        if inout_convs:
            self.in_conv = Conv2d(in_channels, hidden_channels, kernel_size=3, padding="same")
            self.out_conv = Conv2d(hidden_channels, in_channels, kernel_size=3, padding="same")
            width = hidden_channels

        def stage(ch):
            seq = Sequential()
            for _ in range(num_conv_blocks):
                seq.append(ConvBlock(dim=ch))
            return seq

        for _ in range(scales - 1):
            self.conv_sequences.append(stage(width))
            self.downsampling_layers.append(Downsample(in_channels=width))
            width *= rate**2
        self.conv_sequences.append(stage(width))
        for _ in range(scales - 1):
            self.upsampling_layers.append(Upsample(in_channels=width, rate=rate))
            width //= rate**2
            self.conv_sequences.append(stage(width))

    def _stage(self, index, x, inner):
        blocks = self.conv_sequences[index]
        if inner and len(blocks) == 1:
            return blocks[0](x, twice=True)          # x + mlp(x) + xb with xb == x, fused
        xb = x
        for blk in blocks:
            x = blk(x)
        return x + xb if inner else x

    def forward(self, x, x_is_nchw=True):
        """x: NCHW image when x_is_nchw, else NHWC. Returns NCHW when the input was NCHW."""
        x0 = x
        has_io = hasattr(self, "in_conv")
        if has_io:
            x = _ops.Conv3x3Fn.apply(x.contiguous(), self.in_conv.weight, self.in_conv.bias, None, x_is_nchw, False)
        elif x_is_nchw:
            x = x.permute(0, 2, 3, 1).contiguous()
        skips = []
        for lvl in range(self.scales - 1):
            x = self._stage(lvl, x, self.inner_residual)
            skips.append(x)
            x = self.downsampling_layers[lvl](x)
        x = self._stage(self.scales - 1, x, False)
        for lvl in range(self.scales - 1):
            x = self.upsampling_layers[lvl](x, skips.pop())
            x = self._stage(self.scales + lvl, x, False)
        if has_io:
            res = x0.contiguous() if self.residual else None
            return _ops.Conv3x3Fn.apply(x, self.out_conv.weight, self.out_conv.bias, res, False, x_is_nchw)
        if x_is_nchw:
            x = x.permute(0, 3, 1, 2).contiguous()
        return x + x0 if self.residual else x


class ConvolutionalModel(Module):
    def __init__(self, in_channels, upsampling_rate, residual, inner_residual, num_conv_blocks,
                 hidden_channels, inout_convs, scales):
        super().__init__()
        self.seq = Sequential()
        self.scales = scales
        self.upsampling_rate = upsampling_rate
        if upsampling_rate != 1:
            self.seq.append(Upsample(in_channels=in_channels, out_channels=in_channels, rate=upsampling_rate))
        self.seq.append(UNet(in_channels=in_channels, hidden_channels=hidden_channels, inout_convs=inout_convs,
                             scales=scales, num_conv_blocks=num_conv_blocks, residual=residual,
                             inner_residual=inner_residual, rate=2))
        self.flat_params = None
        self.flat_grads = None
        self.flat_shadow = None
        self._sei_plain_state = {"gen": -1, "version": {}}
        # load_state_dict copies into the parameters: cached bf16 shadows are stale afterwards
        self.register_load_state_dict_post_hook(lambda module, incompatible: _ops.weights_updated())

    # -- parameter bucket ----------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.flatten_parameters()
        return out

    def flatten_parameters(self):
        """Re-home every parameter into one contiguous bucket (and prepare the gradient bucket)."""
        params = list(self.parameters())
        if not params:
            return
        # 1x1-convolution weights (the GEMM operands) go last, so that everything else -- the part of the
        # gradient bucket that must be zeroed every step in store mode -- is one contiguous head
        params.sort(key=lambda p: p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1)
        dev, dt = params[0].device, params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in params):
            return
        # every parameter starts on a 256-byte boundary of the bucket (64 float32): 16-byte vector access
        # for all of them whatever the sizes before; the padding stays zero in all three buckets
        align = 64
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        flat = torch.zeros(total, dtype=dt, device=dev)
        grads = torch.zeros(total, dtype=dt, device=dev)
        for p, off in zip(params, offsets):
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p._sei_grad_view = grads[off:off + n].view(p.shape)
            p.grad = None
        self.flat_params, self.flat_grads = flat, grads
        # bf16 copy of the whole bucket for the throughput mode (written by the fused Adam kernel)
        self.flat_shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        # validity of the bf16 bucket is tracked per model: {"gen": generation it was written for,
        # "version": torch version counter of each parameter at that time}
        self._sei_plain_state = {"gen": -1, "version": {}}
        self._sei_zero_ranges = None
        if self.flat_shadow is not None:
            for p, off in zip(params, offsets):
                p._sei_shadow_view = self.flat_shadow[off:off + p.numel()]
                p._sei_plain_state = self._sei_plain_state
                p._sei_shadow = None

    def zero_grad_flat(self, store_weight_grads=False):
        """One memset for the whole model; leaves every p.grad attached to the bucket.

        store_weight_grads=True (GraphedLossStep, after `plan_weight_grad_store`): the 1x1-convolution weight
        gradients are not zeroed -- the step's first launch into each of them stores (models/_ops.py)."""
        ranges = self._sei_zero_ranges if store_weight_grads else None
        if ranges is None:
            self.flat_grads.zero_()
        else:
            for off, n in ranges:
                self.flat_grads[off:off + n].zero_()
        _ops.begin_step(store=ranges is not None)
        for p in self.parameters():
            p.grad = p._sei_grad_view

    def plan_weight_grad_store(self):
        """After at least one eager step: the parts of the gradient bucket that still need zeroing when the
        weight gradients recorded by models/_ops.py are stored rather than accumulated."""
        base, esz = self.flat_grads.data_ptr(), self.flat_grads.element_size()
        total = self.flat_grads.numel()
        skip = sorted(((ptr - base) // esz, n) for ptr, n in _ops.weight_grad_views().items()
                      if base <= ptr < base + total * esz)
        ranges, pos = [], 0
        for off, n in skip:
            if off > pos:
                ranges.append((pos, off - pos))
            pos = max(pos, off + n)
        if pos < total:
            ranges.append((pos, total - pos))
        self._sei_zero_ranges = ranges if skip else None
        return self._sei_zero_ranges

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, y):
        y = N.check_tensor(y.contiguous(), "y")        # crops arrive as strided views
        _ops.note_forward()
        div = 2 ** (self.scales - 1)
        pad_h = (div - y.shape[-2] % div) % div
        pad_w = (div - y.shape[-1] % div) % div
        if pad_h != 0 or pad_w != 0:
            y = F.pad(y, (0, pad_w, 0, pad_h), mode="reflect")
        unet = self.seq[-1]
        if self.upsampling_rate != 1:
            x = self.seq[0](y.permute(0, 2, 3, 1).contiguous())     # NHWC, 3 channels
            x_hat = unet(x, x_is_nchw=False).permute(0, 3, 1, 2)
        else:
            x_hat = unet(y.contiguous(), x_is_nchw=True)
        # crop exactly as the reference does (:296-301): by the INPUT padding, also when the model
        # upsamples (an SR input that needed padding keeps (rate-1)*pad extra rows, as upstream)
        if pad_h != 0 and pad_w != 0:
            x_hat = x_hat[:, :, :-pad_h, :-pad_w]
        elif pad_h != 0:
            x_hat = x_hat[:, :, :-pad_h, :]
        elif pad_w != 0:
            x_hat = x_hat[:, :, :, :-pad_w]
        return x_hat.contiguous()
