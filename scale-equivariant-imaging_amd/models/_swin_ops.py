"""Autograd functions of the SwinIR backbone, each a short sequence of HIP kernel launches (models/swinir.py).

Tokens live as one float32 matrix (B*H*W, C) in natural (b, y, x) order, which is also the NHWC image the 3x3
convolutions read: PatchEmbed / PatchUnEmbed of the reference are views here. Linear layers are the GEMMs of
models/_ops.py (exact-f32 MFMA or bf16 MFMA by `_ops.set_compute_dtype`), window attention is
`sei_swin_attn_fwd/bwd` (shift, window partition, bias lookup and mask inside the kernel), the many-channel 3x3
convolutions are nine row-shifted GEMMs over a zero-bordered copy of the image (`sei_pad_nhwc`): no im2col buffer.
As in models/_ops.py, parameter gradients are accumulated straight into `param.grad` (views of the flat bucket).
"""
import torch

import _native as N
from . import _ops
from ._ops import (EPI_ACCUM, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_NONE, colsum_into_inline as colsum_into, gemm,
                   grad_of)

EPI_BIAS_SCALE_RES = 7
LN_EPS = 1e-5                   # torch.nn.LayerNorm default, which SwinIR uses (the U-Net passes 1e-6)


def layer_norm(x2d, gamma, beta):
    rows, C = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    N.call("sei_ln_fwd", x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, C, LN_EPS)
    return y, mean, rstd


def add(a, b):
    """a + b through the axpy kernel (no torch arithmetic on the path)."""
    out = torch.empty_like(a)
    N.call("sei_axpy", a.data_ptr(), b.data_ptr(), 1.0, out.data_ptr(), a.numel())
    return out


def rowscale(x2d, rows=None, leaky_gate=None):
    y = torch.empty_like(x2d)
    M, Nn = x2d.shape
    N.call("sei_rowscale", x2d.data_ptr(), N.ptr(rows), N.ptr(leaky_gate), y.data_ptr(), M, Nn)
    return y


def window_attention(qkv, table, B, H, W, heads, shift):
    M, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty((M, C), dtype=torch.float32, device=qkv.device)
    N.call("sei_swin_attn_fwd", qkv.data_ptr(), table.data_ptr(), out.data_ptr(), B, H, W, heads, C // heads, shift,
           float((C // heads) ** -0.5))
    return out


def window_attention_bwd(qkv, table, dout, dtable, B, H, W, heads, shift):
    C = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    N.call("sei_swin_attn_bwd", qkv.data_ptr(), table.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(),
           B, H, W, heads, C // heads, shift, float((C // heads) ** -0.5))
    return dqkv


class LayerNormFn(torch.autograd.Function):
    """LayerNorm over the channels of a (B, H, W, C) token image (patch_embed.norm, norm)."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        N.check_tensor(x, "tokens")
        C = x.shape[-1]
        y, mean, rstd = layer_norm(x.view(-1, C), gamma, beta)
        ctx.save_for_backward(x, mean, rstd)
        ctx.params = (gamma, beta)
        return y.view_as(x)

    @staticmethod
    def backward(ctx, go):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        C = x.shape[-1]
        gx = _ops.layer_norm_bwd(x.view(-1, C), gamma, mean, rstd, go.contiguous().view(-1, C), grad_of(gamma),
                                 grad_of(beta))
        return gx.view_as(x), None, None


class SwinBlockFn(torch.autograd.Function):
    """SwinTransformerBlock.forward (network_swinir.py): x + dp(proj(attn(LN1 x))) then + dp(fc2(gelu(fc1(LN2 .)))).
    drop1 / drop2: per-ROW stochastic-depth factors (B*H*W,) or None."""

    @staticmethod
    def forward(ctx, x, g1, b1, table, wqkv, bqkv, wproj, bproj, g2, b2, w1, bm1, w2, bm2, heads, shift, drop1, drop2):
        ctx.dtype = _ops.get_compute_dtype()
        N.check_tensor(x, "tokens")
        B, H, W, C = x.shape
        M, Ch = B * H * W, w1.shape[0]
        x2 = x.view(M, C)
        h1, mean1, rstd1 = layer_norm(x2, g1, b1)
        qkv = gemm(h1, wqkv, M, 3 * C, C, 0, 1, EPI_BIAS, bias=bqkv)
        a = window_attention(qkv, table, B, H, W, heads, shift)
        if drop1 is None:
            x1 = gemm(a, wproj, M, C, C, 0, 1, EPI_BIAS_RES, bias=bproj, R1=x2)
        else:
            x1 = gemm(a, wproj, M, C, C, 0, 1, EPI_BIAS_SCALE_RES, bias=bproj, R1=drop1, R2=x2)
        h2, mean2, rstd2 = layer_norm(x1, g2, b2)
        f4 = torch.empty((M, Ch), dtype=torch.float32, device=x.device)
        f3 = gemm(h2, w1, M, Ch, C, 0, 1, EPI_BIAS_GELU, bias=bm1, D2=f4)
        if drop2 is None:
            out = gemm(f4, w2, M, C, Ch, 0, 1, EPI_BIAS_RES, bias=bm2, R1=x1)
        else:
            out = gemm(f4, w2, M, C, Ch, 0, 1, EPI_BIAS_SCALE_RES, bias=bm2, R1=drop2, R2=x1)
        ctx.save_for_backward(x, mean1, rstd1, h1, qkv, a, x1, mean2, rstd2, h2, f3, f4, drop1, drop2)
        ctx.params = (g1, b1, table, wqkv, bqkv, wproj, bproj, g2, b2, w1, bm1, w2, bm2)
        ctx.cfg = (heads, shift)
        return out.view(B, H, W, C)

    @_ops._in_forward_mode
    def backward(ctx, go):
        x, mean1, rstd1, h1, qkv, a, x1, mean2, rstd2, h2, f3, f4, drop1, drop2 = ctx.saved_tensors
        g1, b1, table, wqkv, bqkv, wproj, bproj, g2, b2, w1, bm1, w2, bm2 = ctx.params
        heads, shift = ctx.cfg
        B, H, W, C = x.shape
        M, Ch = B * H * W, w1.shape[0]
        go2 = go.contiguous().view(M, C)
        # MLP branch
        gy = go2 if drop2 is None else rowscale(go2, drop2)
        colsum_into(grad_of(bm2), gy)
        gemm(gy, f4, C, Ch, M, 1, 0, EPI_ACCUM, out=grad_of(w2))
        gf3 = gemm(gy, w2, M, Ch, C, 0, 0, EPI_MUL_DGELU, R1=f3)
        colsum_into(grad_of(bm1), gf3)
        gemm(gf3, h2, Ch, C, M, 1, 0, EPI_ACCUM, out=grad_of(w1))
        gh2 = gemm(gf3, w1, M, C, Ch, 0, 0, EPI_NONE)
        gx1 = add(_ops.layer_norm_bwd(x1, g2, mean2, rstd2, gh2, grad_of(g2), grad_of(b2)), go2)
        # attention branch
        gy = gx1 if drop1 is None else rowscale(gx1, drop1)
        colsum_into(grad_of(bproj), gy)
        gemm(gy, a, C, C, M, 1, 0, EPI_ACCUM, out=grad_of(wproj))
        ga = gemm(gy, wproj, M, C, C, 0, 0, EPI_NONE)
        dqkv = window_attention_bwd(qkv, table, ga, grad_of(table), B, H, W, heads, shift)
        colsum_into(grad_of(bqkv), dqkv)
        gemm(dqkv, h1, 3 * C, C, M, 1, 0, EPI_ACCUM, out=grad_of(wqkv))
        gh1 = gemm(dqkv, wqkv, M, C, 3 * C, 0, 0, EPI_NONE)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = add(_ops.layer_norm_bwd(x.view(M, C), g1, mean1, rstd1, gh1, grad_of(g1), grad_of(b1)), gx1)
            gx = gx.view(B, H, W, C)
        else:
            _ops.layer_norm_bwd(x.view(M, C), g1, mean1, rstd1, gh1, grad_of(g1), grad_of(b1))
        return (gx,) + (None,) * 17


_TAPS = [(ky, kx) for ky in range(3) for kx in range(3)]


class Conv3x3GemmFn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, 3, 1, 1) on an NHWC batch with many channels (RSTB.conv, conv_after_body,
    conv_before_upsample, upsample.*), then LeakyReLU(0.01) if act, then + res if given.

    The image is copied once onto a zero-bordered grid (B, H+2, W+2, Cin) with guard rows; on that grid output row r
    is sum_taps xp[r + dy*(W+2) + dx] W_tap^T: nine GEMMs whose A operands are row-shifted views of ONE array (the
    first stores with the bias, the rest accumulate), evaluated on the padded grid too (border rows are discarded
    by the un-padding copy). The two gradients have the same form: dW_tap = go_p^T xp[shifted], dxp = sum_taps
    go_p[shifted the other way] W_tap."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act):
        ctx.dtype = _ops.get_compute_dtype()
        N.check_tensor(x, "conv3x3 input")
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        if Cin % 4 or Cout % 4 or tuple(weight.shape) != (Cout, Cin, 3, 3):
            raise ValueError("Conv3x3GemmFn: channel counts must be multiples of 4 and the weight (Cout, Cin, 3, 3)")
        Wp, R = W + 2, B * (H + 2) * (W + 2)
        guard = Wp + 1
        xp = torch.empty((R + 2 * guard, Cin), dtype=torch.float32, device=x.device)
        N.call("sei_pad_nhwc", x.data_ptr(), xp.data_ptr(), B, H, W, Cin, guard)
        wt = weight.detach().permute(2, 3, 0, 1).contiguous().view(9, Cout, Cin)      # tap-major copy (data movement)
        outp = torch.empty((R, Cout), dtype=torch.float32, device=x.device)
        for t, (ky, kx) in enumerate(_TAPS):
            off = guard + (ky - 1) * Wp + (kx - 1)
            gemm(xp[off:off + R], wt[t], R, Cout, Cin, 0, 1, EPI_BIAS if t == 0 else EPI_ACCUM, out=outp, bias=bias,
                 allow_splitk=False)
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        if res is not None:
            N.check_tensor(res, "conv3x3 residual")
        N.call("sei_unpad_nhwc", outp.data_ptr(), N.ptr(res), y.data_ptr(), B, H, W, Cout, int(act))
        ctx.save_for_backward(xp, wt, y if act else None)
        ctx.params, ctx.cfg = (weight, bias), (B, H, W, Cin, Cout, guard, act)
        return y

    @_ops._in_forward_mode
    def backward(ctx, go):
        xp, wt, y_act = ctx.saved_tensors
        weight, bias = ctx.params
        B, H, W, Cin, Cout, guard, act = ctx.cfg
        Wp, R, M = W + 2, B * (H + 2) * (W + 2), B * H * W
        go = go.contiguous()
        gpre = go.view(M, Cout)
        if act:
            gpre = rowscale(gpre, None, leaky_gate=y_act.view(M, Cout))
        colsum_into(grad_of(bias), gpre)
        gop = torch.empty((R + 2 * guard, Cout), dtype=torch.float32, device=go.device)
        N.call("sei_pad_nhwc", gpre.data_ptr(), gop.data_ptr(), B, H, W, Cout, guard)
        g0 = gop[guard:guard + R]
        dwt = torch.zeros((9, Cout, Cin), dtype=torch.float32, device=go.device)
        for t, (ky, kx) in enumerate(_TAPS):
            off = guard + (ky - 1) * Wp + (kx - 1)
            gemm(g0, xp[off:off + R], Cout, Cin, R, 1, 0, EPI_ACCUM, out=dwt[t])
        grad_of(weight).add_(dwt.view(3, 3, Cout, Cin).permute(2, 3, 0, 1))
        gx = None
        if ctx.needs_input_grad[0]:
            dxp = torch.empty((R, Cin), dtype=torch.float32, device=go.device)
            for t, (ky, kx) in enumerate(_TAPS):
                off = guard - (ky - 1) * Wp - (kx - 1)
                gemm(gop[off:off + R], wt[t], R, Cin, Cout, 0, 0, EPI_NONE if t == 0 else EPI_ACCUM, out=dxp,
                     allow_splitk=False)
            gx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=go.device)
            N.call("sei_unpad_nhwc", dxp.data_ptr(), None, gx.data_ptr(), B, H, W, Cin, 0)
        gres = go if ctx.needs_input_grad[3] else None
        return gx, None, None, gres, None
