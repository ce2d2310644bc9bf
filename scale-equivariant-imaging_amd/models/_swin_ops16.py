"""SwinIR in throughput (bf16) mode: the same blocks as models/_swin_ops.py on the LDS-DMA bf16 GEMMs
(`sei_gemm_bf16nt`), the MFMA window attention (`sei_swin_attn_*_bf16`) and the implicit-GEMM 3x3 convolution
(`sei_gemm_bf16nt_conv`).

Layouts. Tokens stay float32 (B*H*W, 180) between blocks. Every GEMM operand is bf16 with its reduction dimension
padded to a multiple of 64 and exact zeros in the pad: LayerNorm / cast kernels write (M, 192) rows, attention heads
are 32 wide (30 + 2 zeros: q | k | v of all heads = 576 columns), convolution inputs live on the zero-bordered
grid with 192 (or 64 / 256) channels. The weights are re-laid out to match by ONE gather kernel per model call
(`SwinPack.refresh`: float32 bucket -> bf16 GEMM layouts through an index map), and the weight gradients, which the
GEMMs accumulate in those layouts, return to the flat gradient bucket through the same map once per backward pass
(`SwinPack.flush`).
"""
import numpy as np
import torch

import _native as N
from . import _ops
from ._ops import (EPI_ACCUM, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_MUL_DGELU, EPI_NONE, colsum16_into_inline as colsum16_into,
                   colsum_into_inline as colsum_into,
                   gemm_nt16, grad_of, weight_grad16, weight_grad16_group)
from ._swin_ops import EPI_BIAS_SCALE_RES, LN_EPS, rowscale

CP, HP = 192, 32                # padded embedding width, padded head width


def pad64(n):
    return (n + 63) // 64 * 64


class SwinPack:
    """bf16 GEMM-layout copies of a SwinIR's matrices and float32 staging for their gradients."""

    def __init__(self, model):
        flat = model.flat_params
        if flat is None or not flat.is_cuda:
            raise ValueError("SwinPack needs a SwinIR whose parameters live in the flat bucket on the GPU")
        self.model = model
        base, esz = flat.data_ptr(), flat.element_size()
        wmaps, gmaps, bmaps = [], [], []
        self._w, self._g, self._b = {}, {}, {}
        self.bias_in_taps = set()                       # convolutions whose bias gradient is a column of their weight gradient
        cursor = {"w": 0, "g": 0, "b": 0}

        def index_of(p):                                # flat-bucket index of every element of p, shaped like p
            off = (p.data_ptr() - base) // esz
            return off + np.arange(p.numel(), dtype=np.int64).reshape(tuple(p.shape))

        def add(kind, key, imap):
            store, maps = {"w": (self._w, wmaps), "g": (self._g, gmaps), "b": (self._b, bmaps)}[kind]
            n = imap.size
            n_pad = (n + 63) // 64 * 64                 # every block starts on a 128-byte (bf16) boundary
            store[key] = (cursor[kind], imap.shape)
            maps.append(np.concatenate([imap.reshape(-1), np.full(n_pad - n, -1, dtype=np.int64)]))
            cursor[kind] += n_pad

        heads = model.blocks()[0].num_heads
        C = model.embed_dim
        hd = C // heads
        if (C, hd) != (180, 30):
            raise NotImplementedError("the bf16 SwinIR path is laid out for embed_dim 180 / head_dim 30")
        head_cols = np.full(heads * HP, -1, dtype=np.int64)              # padded head column -> real column
        for h in range(heads):
            head_cols[h * HP:h * HP + hd] = h * hd + np.arange(hd)

        def pad_cols(m, width):                          # (r, c) -> (r, width), -1 beyond c
            out = np.full((m.shape[0], width), -1, dtype=np.int64)
            out[:, :m.shape[1]] = m
            return out

        def pad_rows(m, height):
            out = np.full((height, m.shape[1]), -1, dtype=np.int64)
            out[:m.shape[0]] = m
            return out

        for name, blk in model.named_modules():
            if type(blk).__name__ != "SwinTransformerBlock":
                continue
            a, mlp = blk.attn, blk.mlp
            wq = index_of(a.qkv.weight)                  # (540, 180)
            rows = np.concatenate([s * C + head_cols for s in range(3)])          # padded qkv row -> real row (or junk)
            rows = np.where(np.tile(head_cols, 3) >= 0, rows, -1)
            mq = np.where(rows[:, None] >= 0, pad_cols(wq, CP)[np.maximum(rows, 0)], -1)
            bq = np.where(rows >= 0, index_of(a.qkv.bias)[np.maximum(rows, 0)], -1)
            wp = index_of(a.proj.weight)                 # (180, 180): columns follow the padded heads
            mp = np.where(head_cols[None, :] >= 0, wp[:, np.maximum(head_cols, 0)], -1)
            hid = mlp.fc1.weight.shape[0]
            hidp = pad64(hid)                            # the hidden activations are (M, 384) rows: whole 64-column k-tiles
            layouts = {"qkv": mq, "proj": pad_rows(mp, CP), "fc1": pad_rows(pad_cols(index_of(mlp.fc1.weight), CP), hidp),
                       "fc2": pad_cols(pad_rows(index_of(mlp.fc2.weight), CP), hidp)}
            # gradient layouts: as the weights, plus -- for the two layers fed by a LayerNorm, whose padded rows carry a
            # column of ones (ln16) -- the bias in column C: dY^T [h | 1] puts the bias gradient there
            grads = dict(layouts)
            grads["qkv"] = mq.copy()
            grads["qkv"][:, C] = bq
            grads["fc1"] = layouts["fc1"].copy()
            grads["fc1"][:hid, C] = index_of(mlp.fc1.bias)
            # fc2: the gelu output's first padding column holds 1.0 where fc1 ran on sei_rowgemm_gelu_bf16 (0.0 otherwise)
            grads["fc2"] = layouts["fc2"].copy()
            grads["fc2"][:C, hid] = index_of(mlp.fc2.bias)
            for k, m in layouts.items():
                add("w", f"{name}.{k}", m)
                add("w", f"{name}.{k}T", np.ascontiguousarray(m.T))     # the data gradients read the transposes K-contiguous
                add("g", f"{name}.{k}", grads[k])
            add("b", f"{name}.qkv_bias", bq)
            b1 = np.full(hidp, -1, dtype=np.int64)
            b1[:hid] = index_of(mlp.fc1.bias)
            add("b", f"{name}.fc1_bias", b1)
        for name, conv in model.named_modules():
            if not isinstance(conv, torch.nn.Conv2d):
                continue
            w = index_of(conv.weight)                    # (Cout, Cin, 3, 3)
            Cout, Cin = w.shape[:2]
            cin4, cout4 = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4          # activations carry whole float4 pixels
            cinp, coutp = pad64(cin4), pad64(cout4)
            fwd = np.full((cout4, 9, cinp), -1, dtype=np.int64)
            fwd[:Cout, :, :Cin] = w.reshape(Cout, Cin, 9).transpose(0, 2, 1)
            bwd = np.full((cin4, 9, coutp), -1, dtype=np.int64)
            bwd[:Cin, :, :Cout] = w.reshape(Cout, Cin, 9).transpose(1, 2, 0)
            grd = np.full((9, coutp, cinp), -1, dtype=np.int64)
            grd[:, :Cout, :Cin] = w.reshape(Cout, Cin, 9).transpose(2, 0, 1)
            if cin4 < cinp and conv.bias is not None and Cout % 4 == 0:
                # a column of ones in the padded input grid (channel cin4: sei_pad_nhwc_bf16_ones) puts the bias gradient
                # into that column of every tap's weight gradient; the centre tap's copy is routed to the bias
                grd[4, :Cout, cin4] = index_of(conv.bias)
                self.bias_in_taps.add(name)
            add("w", f"{name}.fwd", fwd.reshape(cout4, 9 * cinp))
            add("w", f"{name}.bwd", bwd.reshape(cin4, 9 * coutp))
            add("g", f"{name}.taps", grd)
            if Cout % 4:                                 # conv_last: 3 outputs computed as 4 (the 4th is zero)
                b4 = np.full(cout4, -1, dtype=np.int64)
                b4[:Cout] = index_of(conv.bias)
                add("b", f"{name}.bias4", b4)
                add("g", f"{name}.bias4", b4)
        dev = flat.device

        def to_dev(maps):
            return torch.from_numpy(np.concatenate(maps).astype(np.int32)).to(dev)

        self.wmap, self.gmap, self.bmap = to_dev(wmaps), to_dev(gmaps), to_dev(bmaps)
        self.wint = torch.zeros(self.wmap.numel(), dtype=torch.bfloat16, device=dev)
        self.bint = torch.zeros(self.bmap.numel(), dtype=torch.float32, device=dev)
        self.gint = torch.zeros(self.gmap.numel(), dtype=torch.float32, device=dev)
        _ops.register_gradient_range(model, self.gint)  # these staged gradients are part of `model`'s step bookkeeping
        self._flush_queued = False
        self._key = (base, flat.numel())

    def valid_for(self, model):
        flat = model.flat_params
        return flat is not None and self._key == (flat.data_ptr(), flat.numel())

    def refresh(self):
        """bf16 GEMM layouts <- the float32 parameters (two gather launches; call before every forward)."""
        flat = self.model.flat_params
        N.call("sei_pack", flat.data_ptr(), self.wmap.data_ptr(), self.wint.data_ptr(), self.wint.numel(), 1)
        N.call("sei_pack", flat.data_ptr(), self.bmap.data_ptr(), self.bint.data_ptr(), self.bint.numel(), 0)

    def w(self, key):
        off, shape = self._w[key]
        return self.wint[off:off + int(np.prod(shape))].view(shape)

    def b(self, key):
        off, shape = self._b[key]
        return self.bint[off:off + int(np.prod(shape))].view(shape)

    def g(self, key):
        """float32 gradient staging of `key` (GEMM-output layout); arms the end-of-backward flush."""
        if not self._flush_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
                self._flush_queued = True
            except RuntimeError:                         # not inside a backward pass: the caller flushes
                pass
        off, shape = self._g[key]
        return self.gint[off:off + int(np.prod(shape))].view(shape)

    def flush(self):
        """Staged gradients -> the flat gradient bucket (+=), then clear the staging for the next backward pass."""
        self._flush_queued = False
        _ops.flush_weight_grads(self.model)                        # pairs still parked for a partner that never came
        grads = self.model.flat_grads
        for p in self.model.parameters():                # gradients not attached yet (no zero_grad_flat): attach
            if p.grad is None:
                grad_of(p)
        N.call("sei_unpack_add", self.gint.data_ptr(), self.gmap.data_ptr(), grads.data_ptr(), self.gint.numel())
        self.gint.zero_()


def _partials(C, device):
    """Scratch for the per-workgroup partial sums of the two reducing kernels."""
    return torch.empty(N.lib().sei_swin_partials_floats(C), dtype=torch.float32, device=device)


def ln16(x2d, gamma, beta):
    rows, C = x2d.shape
    y = torch.empty((rows, CP), dtype=torch.bfloat16, device=x2d.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    # column C of the padded rows holds 1.0: the weight gradient of the linear layer behind (gy^T [h | 1]) then has that
    # layer's bias gradient in its column C (SwinPack maps it onto the bias); the weights' own padding columns are zero
    N.call("sei_ln_fwd_bf16_pad", x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), rows, C, CP, LN_EPS, 1)
    return y, mean, rstd


def ln_bwd(x2d, gamma, mean, rstd, gy, res, ggamma, gbeta):
    rows, C = x2d.shape
    gx = torch.empty_like(x2d)
    work = _partials(C, x2d.device)
    N.call("sei_ln_bwd_pad", x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(),
           N.ptr(res), gx.data_ptr(), ggamma.data_ptr(), gbeta.data_ptr(), rows, C, gy.shape[1], work.data_ptr(),
           work.numel())
    return gx


def cast_pad(x2d, rows_scale=None, colsum=None, width=CP):
    M, C = x2d.shape
    y = torch.empty((M, width), dtype=torch.bfloat16, device=x2d.device)
    work = _partials(C, x2d.device) if colsum is not None else None
    N.call("sei_cast_pad_bf16", x2d.data_ptr(), N.ptr(rows_scale), y.data_ptr(), N.ptr(colsum), M, C, width,
           N.ptr(work), 0 if work is None else work.numel())
    return y


def linear16(A16, W, Wt, M, epi, nv, out32=None, out16=None, bias=None, R1=None, R2=None, flops=None):
    """D = epilogue(A16 W^T) for a layer-sized W (N, K): the row-streaming kernel (sei_rowgemm_bf16) where it is built
    for the shape, the tiled GEMM otherwise. Wt: what the tiled kernel reads instead for a data gradient (the forward
    matrix (K, N), reduction-major); None for a forward product."""
    Nn, K = W.shape
    only16 = out16 is not None and out32 is None
    if _ops.TOKEN_STREAMING and N.lib().sei_rowgemm_bf16_eligible(M, Nn, K, epi, int(only16)):
        rows = R2 if epi == EPI_BIAS_SCALE_RES else R1
        _ops._gemm_call(2.0 * M * Nn * K if flops is None else float(flops), "sei_rowgemm_bf16", A16.data_ptr(), A16.shape[1],
                        W.data_ptr(), K, N.ptr(out32), 0 if out32 is None else out32.shape[1], N.ptr(out16),
                        0 if out16 is None else out16.shape[1], M, Nn, K, nv, epi, N.ptr(bias), N.ptr(R1), N.ptr(R2),
                        0 if rows is None else rows.shape[1])
        return
    if epi == EPI_BIAS_GELU:
        gemm_nt16(A16, W, M, nv, K, epi, out32=out32, bias=bias, D2_16=out16, flops=flops)
    elif Wt is None:
        gemm_nt16(A16, W, M, nv, K, epi, out32=out32, out16=out16, bias=bias, R1=R1, R2=R2, flops=flops)
    else:
        gemm_nt16(A16, Wt, M, nv, K, epi, out32=out32, out16=out16, R1=R1, b_rmajor=True, flops=flops)


def linear_lnbwd16(A16, W, Wt, M, x2d, gamma, mean, rstd, res, ggamma, gbeta, row_scale=None, colsum=None, cast=None):
    """gx = LayerNorm backward of gh = A16 W^T (the data gradient of the linear layer behind the norm) + res, the norm's
    weight / bias gradients accumulated, and -- with row_scale / colsum -- gy16 = bf16(gx * row_scale) in padded rows with
    its column sums added to colsum: one launch where the fused kernel is built (sei_rowgemm_lnbwd_bf16), GEMM +
    sei_ln_bwd_pad + sei_cast_pad_bf16 otherwise. Returns (gx, gy16 or None)."""
    Nn, K = W.shape
    C = x2d.shape[1]
    want16 = colsum is not None or bool(cast)            # cast: the scaled bf16 copy without column sums (any K)
    if (_ops.TOKEN_STREAMING and Nn == CP and N.lib().sei_rowgemm_lnbwd_bf16_eligible(M, K, C)
            and (not want16 or row_scale is not None) and (colsum is None or K == 384)):
        gx = torch.empty_like(x2d)
        gy16 = torch.empty((M, CP), dtype=torch.bfloat16, device=x2d.device) if want16 else None
        work = torch.empty(N.lib().sei_rowgemm_lnbwd_work_floats(C), dtype=torch.float32, device=x2d.device)
        deferred = _ops.defer_fold(ggamma, gbeta, colsum, 3 * C, C, N.FOLD_SPLIT, work, 0, min(M // 32, 256))
        _ops._gemm_call(2.0 * M * C * K, "sei_rowgemm_lnbwd_bf16", A16.data_ptr(), A16.shape[1], W.data_ptr(), K, M, K,
                        x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), res.data_ptr(), gx.data_ptr(), C,
                        None if deferred else ggamma.data_ptr(), None if deferred else gbeta.data_ptr(),
                        N.ptr(row_scale) if want16 else None, N.ptr(gy16), CP, N.ptr(colsum), work.data_ptr(), work.numel())
        return gx, gy16
    gh = torch.empty((M, CP), dtype=torch.float32, device=x2d.device)
    linear16(A16, W, Wt, M, EPI_NONE, CP, out32=gh, flops=2.0 * M * C * K)
    gx = ln_bwd(x2d, gamma, mean, rstd, gh, res, ggamma, gbeta)
    return gx, (cast_pad(gx, row_scale, colsum) if want16 else None)


def linear_res_ln16(A16, W, M, bias, drop, res, gamma, beta, flops):
    """out = res + [drop] (A16 W^T + bias) and the LayerNorm (gamma, beta) of out as padded bf16 rows with its statistics:
    one launch (sei_rowgemm_ln_bf16) where built, else None (the caller runs the layer and the norm separately)."""
    K = W.shape[1]
    C = res.shape[1]
    if not (_ops.TOKEN_STREAMING and N.lib().sei_rowgemm_ln_bf16_eligible(M, K, C)):
        return None
    dev = res.device
    out = torch.empty((M, C), dtype=torch.float32, device=dev)
    h = torch.empty((M, CP), dtype=torch.bfloat16, device=dev)
    mean = torch.empty(M, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    _ops._gemm_call(flops, "sei_rowgemm_ln_bf16", A16.data_ptr(), A16.shape[1], W.data_ptr(), K, M, K, C, bias.data_ptr(),
                    N.ptr(drop), res.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), LN_EPS, 1, h.data_ptr(), CP,
                    mean.data_ptr(), rstd.data_ptr())
    return out, h, mean, rstd


class SwinBlockFn16(torch.autograd.Function):
    """models._swin_ops.SwinBlockFn in bf16 mode; `pack` / `key` give the block's re-laid-out matrices."""

    @staticmethod
    def forward(ctx, x, g1, b1, table, bproj, g2, b2, bm1, bm2, pack, key, heads, shift, drop1, drop2, pre=None, nxt=None,
                prev_scale=None):
        """pre: (h1, mean1, rstd1) = this block's norm1 already applied to x by the layer that produced x (the previous
        block's fc2 launch); nxt: (weight, bias) of the NEXT block's norm1 -- the block then returns (out, h, mean, rstd)
        with that norm applied to its output (None entries where the fused launch is not built for the shape). prev_scale:
        the per-row stochastic-depth factors of the PREVIOUS block's MLP branch (the block whose output x is): this block's
        norm1 backward then also leaves bf16(gx * prev_scale) -- what the previous block's backward would cast first."""
        N.check_tensor(x, "tokens")
        B, H, W, C = x.shape
        M = B * H * W
        x2 = x.view(M, C)
        wqkv, wproj, w1, w2 = (pack.w(f"{key}.{k}") for k in ("qkv", "proj", "fc1", "fc2"))
        Ch, Chr = w1.shape[0], bm1.shape[0]              # padded (384) / real (360) hidden width
        dev = x.device
        h1, mean1, rstd1 = pre if pre is not None else ln16(x2, g1, b1)
        qkv = torch.empty((M, 3 * heads * HP), dtype=torch.bfloat16, device=dev)
        linear16(h1, wqkv, None, M, EPI_BIAS, 3 * heads * HP, out16=qkv, bias=pack.b(f"{key}.qkv_bias"), flops=2.0 * M * 3 * C * C)
        a = torch.empty((M, CP), dtype=torch.bfloat16, device=dev)
        scale = float((C // heads) ** -0.5)
        # per (head, window, query) log-sum-exp of the scores: the backward pass rebuilds the probabilities from it
        lse = torch.empty((heads, M), dtype=torch.float32, device=dev)
        N.call("sei_swin_attn_fwd_bf16", qkv.data_ptr(), table.data_ptr(), a.data_ptr(), lse.data_ptr(), B, H, W, heads, shift,
               scale)
        fused = linear_res_ln16(a, wproj, M, bproj, drop1, x2, g2, b2, 2.0 * M * C * C)    # proj, residual, norm2
        if fused is not None:
            x1, h2, mean2, rstd2 = fused
        else:
            x1 = torch.empty((M, C), dtype=torch.float32, device=dev)
            if drop1 is None:
                linear16(a, wproj, None, M, EPI_BIAS_RES, C, out32=x1, bias=bproj, R1=x2, flops=2.0 * M * C * C)
            else:
                linear16(a, wproj, None, M, EPI_BIAS_SCALE_RES, C, out32=x1, bias=bproj, R1=drop1, R2=x2, flops=2.0 * M * C * C)
            h2, mean2, rstd2 = ln16(x1, g2, b2)
        # the float32 pre-activation is stored only where the backward cannot recompute it (sei_rowgemm_dgelu_bf16)
        recompute = _ops.TOKEN_STREAMING and bool(N.lib().sei_rowgemm_dgelu_bf16_eligible(M, Ch, CP)) and \
            bool(N.lib().sei_rowgemm_bf16_eligible(M, Ch, CP, EPI_BIAS_GELU, 0))
        f3 = None if recompute else torch.empty((M, Ch), dtype=torch.float32, device=dev)
        f4 = torch.empty((M, Ch), dtype=torch.bfloat16, device=dev)
        if recompute:                                    # column Chr of f4 = 1.0: fc2's bias gradient rides in its weight gradient
            _ops._gemm_call(2.0 * M * Chr * C, "sei_rowgemm_gelu_bf16", h2.data_ptr(), CP, w1.data_ptr(), CP,
                            pack.b(f"{key}.fc1_bias").data_ptr(), Ch, f4.data_ptr(), Ch, M, Ch, CP, Chr)
        else:
            linear16(h2, w1, None, M, EPI_BIAS_GELU, Ch, out32=f3, out16=f4, bias=pack.b(f"{key}.fc1_bias"),
                     flops=2.0 * M * Chr * C)
        fused = linear_res_ln16(f4, w2, M, bm2, drop2, x1, nxt[0], nxt[1], 2.0 * M * Chr * C) if nxt is not None else None
        if fused is not None:                            # fc2, residual, the next block's norm1
            out, hn, mean_n, rstd_n = fused
        else:
            hn = mean_n = rstd_n = None
            out = torch.empty((M, C), dtype=torch.float32, device=dev)
            if drop2 is None:
                linear16(f4, w2, None, M, EPI_BIAS_RES, C, out32=out, bias=bm2, R1=x1, flops=2.0 * M * Chr * C)
            else:
                linear16(f4, w2, None, M, EPI_BIAS_SCALE_RES, C, out32=out, bias=bm2, R1=drop2, R2=x1, flops=2.0 * M * Chr * C)
        ctx.save_for_backward(x, mean1, rstd1, h1, qkv, a, x1, mean2, rstd2, h2, f3, f4, drop1, drop2, lse)
        ctx.params = (g1, b1, table, bproj, g2, b2, bm1, bm2)
        ctx.cfg = (pack, key, heads, shift)
        ctx.extra_outputs = nxt is not None
        ctx.ones = recompute                             # f4 carries the ones column
        ctx.prev_scale = prev_scale if recompute else None      # (same token count, same mode: the previous block does too)
        if nxt is None:
            return out.view(B, H, W, C)
        if hn is None:                                   # not fused: the next block's norm in its own launch, here
            hn, mean_n, rstd_n = ln16(out, nxt[0], nxt[1])
        ctx.mark_non_differentiable(hn, mean_n, rstd_n)  # their gradient flows through `out` in the next block's backward
        ctx.set_materialize_grads(False)                 # (no zero tensors made for them in the backward pass)
        return out.view(B, H, W, C), hn, mean_n, rstd_n

    @staticmethod
    def backward(ctx, go, *unused):
        x, mean1, rstd1, h1, qkv, a, x1, mean2, rstd2, h2, f3, f4, drop1, drop2, lse = ctx.saved_tensors
        g1, b1, table, bproj, g2, b2, bm1, bm2 = ctx.params
        pack, key, heads, shift = ctx.cfg
        B, H, W, C = x.shape
        M = B * H * W
        dev = x.device
        wqkv, wproj, w1, w2 = (pack.w(f"{key}.{k}") for k in ("qkv", "proj", "fc1", "fc2"))
        Ch, Chr = w1.shape[0], bm1.shape[0]
        go2 = go.contiguous().view(M, C)
        # MLP branch. The bf16 operand gy = bf16(go * drop2): already formed by the next block's norm1 backward (below,
        # handed over on the gradient tensor itself), else cast here; fc2's bias gradient = its column sums, or -- with the
        # ones column in f4 -- column Chr of fc2's weight gradient
        # (accepted only while `go` is still the very tensor, unmodified, that the next block's backward returned: a second
        # consumer of the block output or an in-place hook makes autograd accumulate into / rewrite it)
        tag = getattr(go, "_sei_cast16", None) if ctx.ones else None
        gy = tag[0] if tag is not None and tag[1] == go.data_ptr() and tag[2] == go._version else None
        if gy is None:
            gy = cast_pad(go2, drop2, None if ctx.ones else grad_of(bm2))
        gf3 = torch.empty((M, Ch), dtype=torch.bfloat16, device=dev)
        if f3 is None:                                   # GELU' of the recomputed pre-activation h2 W1^T + b1
            _ops._gemm_call(4.0 * M * Chr * C, "sei_rowgemm_dgelu_bf16", gy.data_ptr(), CP, pack.w(f"{key}.fc2T").data_ptr(), CP,
                            h2.data_ptr(), CP, w1.data_ptr(), CP, pack.b(f"{key}.fc1_bias").data_ptr(), Ch, gf3.data_ptr(), Ch,
                            M, Ch, CP)
        else:
            linear16(gy, pack.w(f"{key}.fc2T"), w2, M, EPI_MUL_DGELU, Ch, out16=gf3, R1=f3, flops=2.0 * M * Chr * C)
        # fc1's data gradient, norm2's backward (+ the block's incoming gradient), and the cast / stochastic-depth scale /
        # bias column sums that open the attention branch
        gx1, gy1 = linear_lnbwd16(gf3, pack.w(f"{key}.fc1T"), w1, M, x1, g2, mean2, rstd2, go2, grad_of(g2), grad_of(b2),
                                  drop1, grad_of(bproj))
        ga = torch.empty((M, CP), dtype=torch.bfloat16, device=dev)
        linear16(gy1, pack.w(f"{key}.projT"), wproj, M, EPI_NONE, CP, out16=ga, flops=2.0 * M * C * C)
        dqkv = torch.empty_like(qkv)
        scale = float((C // heads) ** -0.5)
        N.call("sei_swin_attn_bwd_bf16", qkv.data_ptr(), table.data_ptr(), a.data_ptr(), lse.data_ptr(), ga.data_ptr(),
               dqkv.data_ptr(), grad_of(table).data_ptr(), B, H, W, heads, shift, scale)
        gx, gy_prev = linear_lnbwd16(dqkv, pack.w(f"{key}.qkvT"), wqkv, M, x.view(M, C), g1, mean1, rstd1, gx1, grad_of(g1),
                                     grad_of(b1), row_scale=ctx.prev_scale, cast=ctx.prev_scale is not None)
        # the four weight gradients (+ the qkv / fc1 bias gradients, column C) over the same tokens: one launch
        weight_grad16_group([(dqkv, h1, pack.g(f"{key}.qkv"), 2.0 * 3 * C * C), (gy1, a, pack.g(f"{key}.proj"), 2.0 * C * C),
                             (gf3, h2, pack.g(f"{key}.fc1"), 2.0 * Chr * C), (gy, f4, pack.g(f"{key}.fc2"), 2.0 * Chr * C)])
        if not ctx.needs_input_grad[0]:
            return (None,) * 18
        gxv = gx.view(B, H, W, C)
        if gy_prev is not None:
            # for the previous block's backward (the same tensor object arrives there), with what identifies its contents
            gxv._sei_cast16 = (gy_prev, gxv.data_ptr(), gxv._version)
        return (gxv,) + (None,) * 17


_TAPS = [(ky, kx) for ky in range(3) for kx in range(3)]
_OFFSETS = {}


def _tap_offsets(Wp, sign):
    key = (Wp, sign)
    if key not in _OFFSETS:
        import ctypes
        _OFFSETS[key] = (ctypes.c_int * 9)(*[sign * ((ky - 1) * Wp + (kx - 1)) for ky, kx in _TAPS])
    return _OFFSETS[key]


class Conv3x3GemmFn16(torch.autograd.Function):
    """models._swin_ops.Conv3x3GemmFn in bf16 mode: ONE implicit GEMM per convolution on the zero-bordered bf16 grid
    (k-tiles = 64-channel slices of one tap, read from the same array at that tap's row shift), the same for the
    data gradient with the transposed tap-major weights, and nine reduction-major GEMMs for the weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, act, pack, key):
        N.check_tensor(x, "conv3x3 input")
        B, H, W, Cin = x.shape                           # Cin, Cout: rounded up to whole float4 pixels (conv_first: the
        wf = pack.w(f"{key}.fwd")                        # image arrives with a zero 4th channel; conv_last: 4th output 0)
        Cout = wf.shape[0]
        if (weight.shape[0] + 3) // 4 * 4 != Cout or (weight.shape[1] + 3) // 4 * 4 != Cin:
            raise ValueError("Conv3x3GemmFn16: activation channels do not match the packed weight")
        if Cout != weight.shape[0]:
            bias = pack.b(f"{key}.bias4")
        cinp = pad64(Cin)
        Wp, R = W + 2, B * (H + 2) * (W + 2)
        guard = Wp + 9                                   # tap shifts (<= Wp + 1) + rounding the row count up to 8
        xp = torch.empty((R + 2 * guard, cinp), dtype=torch.bfloat16, device=x.device)
        ones = key in pack.bias_in_taps                 # the bias gradient rides in the weight gradient (SwinPack)
        N.call("sei_pad_nhwc_bf16_ones", x.data_ptr(), xp.data_ptr(), B, H, W, Cin, cinp, guard, int(ones))
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        if res is not None:
            N.check_tensor(res, "conv3x3 residual")
        # one implicit GEMM whose epilogue drops the border pixels, applies the activation and adds the residual
        _ops._gemm_call(2.0 * B * H * W * weight.shape[0] * 9 * weight.shape[1], "sei_gemm_bf16nt_conv_unpad",
                        xp[guard:].data_ptr(), cinp, _tap_offsets(Wp, 1), wf.data_ptr(), wf.shape[1], y.data_ptr(), N.ptr(res),
                        B, H, W, Cout, bias.data_ptr(), int(act))
        ctx.save_for_backward(xp, y if act else None)
        ctx.params, ctx.cfg = (weight, bias), (B, H, W, Cin, Cout, guard, act, pack, key)
        return y

    @staticmethod
    def backward(ctx, go):
        xp, y_act = ctx.saved_tensors
        weight, bias = ctx.params
        B, H, W, Cin, Cout, guard, act, pack, key = ctx.cfg
        cinp, coutp = pad64(Cin), pad64(Cout)
        Wp, R, M = W + 2, B * (H + 2) * (W + 2), B * H * W
        R8 = (R + 7) // 8 * 8
        go = go.contiguous()
        gpre = go.view(M, Cout)
        if act:
            gpre = rowscale(gpre, None, leaky_gate=y_act.view(M, Cout))
        if key not in pack.bias_in_taps:
            colsum_into(grad_of(bias) if Cout == weight.shape[0] else pack.g(f"{key}.bias4"), gpre)
        gop = torch.empty((R + 2 * guard, coutp), dtype=torch.bfloat16, device=go.device)
        N.call("sei_pad_nhwc_bf16", gpre.data_ptr(), gop.data_ptr(), B, H, W, Cout, coutp, guard)
        taps = pack.g(f"{key}.taps")                     # (9, coutp, cinp): one launch for the nine taps
        weight_grad16(gop[guard:guard + R8], xp[guard:guard + R8], taps,
                      flops_per_row=2.0 * M * weight.shape[0] * 9 * weight.shape[1] / R8, tap_rows=_tap_offsets(Wp, 1))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=go.device)
            wb = pack.w(f"{key}.bwd")
            _ops._gemm_call(2.0 * M * weight.shape[0] * 9 * weight.shape[1], "sei_gemm_bf16nt_conv_unpad", gop[guard:].data_ptr(),
                            coutp, _tap_offsets(Wp, -1), wb.data_ptr(), wb.shape[1], gx.data_ptr(), None, B, H, W, Cin, None, 0)
        gres = go if ctx.needs_input_grad[3] else None
        return gx, None, None, gres, None, None, None
