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
from . import _joint, _mats, _ops
from ._flat import FlatParameterBucket


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

    def forward(self, x, with_skip=False):
        """with_skip: returns (downsampled, x) -- the second output is x handed on for the U-Net's skip connection, so that
        both gradients of x meet inside this layer's backward (models/_ops.DownsampleFn)."""
        fn = _ops.DownsampleFn16 if _ops.use_bf16_blocks(x.shape[-1]) else _ops.DownsampleFn
        return fn.apply(x, self.ln.ln.weight, self.ln.ln.bias, self.conv.weight, self.conv.bias, self.rate, with_skip)


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
        if not inner:
            return x
        # A torch add between taped layer functions: the tape (models/_joint.py) is a chain of layer functions and would
        # drop the gradient that bypasses the blocks through xb -- this call stays an ordinary autograd graph.
        _ops._no_joint_form()
        return x + xb

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
            x, skip = self.downsampling_layers[lvl](x, with_skip=True)     # (reference :226-232: skips.append(x); down(x))
            skips.append(skip)
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


class ConvolutionalModel(FlatParameterBucket, Module):
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
        self._init_bucket()

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, y):
        with _ops.compute_dtype_scope(self):           # this model's own arithmetic mode, if it carries one
            return self._forward(y)

    def _forward(self, y):
        y = N.check_tensor(y.contiguous(), "y")        # crops arrive as strided views
        _ops.note_forward(self)
        div = 2 ** (self.scales - 1)
        pad_h = (div - y.shape[-2] % div) % div
        pad_w = (div - y.shape[-1] % div) % div
        # The step's two model calls (2B and B images, the second input a constant: ProposedLoss with stop_gradient) share
        # ONE backward pass: this call is recorded -- activations from an arena, layer functions on a tape -- when the loss
        # has announced the pair, the mode is bf16 and nothing around the U-Net needs torch ops (no padding, no pre-upsampler)
        rec = _joint.recorder_of(self)
        if rec.armed and self.upsampling_rate == 1 and pad_h == 0 and pad_w == 0 and hasattr(self.seq[-1], "in_conv") \
                and _ops.get_compute_dtype() == "bf16" and rec.wants(y):
            rec.begin(y)
            with _ops.recording(rec):
                y_in = N.copy_into(_ops._alloc(tuple(y.shape), y.dtype, y.device), y)
                x_hat = self.seq[-1](y_in, x_is_nchw=True)
            call, pair = rec.current, rec.pair
            if not rec.end():
                return x_hat.contiguous()               # (no joint form: an ordinary autograd graph)
            pair.outputs[call] = x_hat                  # (its grad_fn keeps the layer nodes -- the tape's ctx objects -- alive)
            return _joint._Top.apply(x_hat.detach(), self.seq[-1].out_conv.bias, pair, call, self)
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
