"""GPU parity: U-Net kernels, layers and whole model vs the oracle and the reference goldens.

Bar (north_star): restored images within 1e-4 relative (max-norm) of the float32 CPU path. The
goldens here are float64 runs of the reference modules, so the measured error is the product's own
float32 rounding; per-kernel tolerances are tighter (1e-5 .. 1e-6).
"""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_path as tp

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def ops():
    from models import _ops
    return _ops


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(288, 512, 128), (2304, 128, 32), (64, 64, 16), (37, 29, 19), (1, 3, 3),
                                    (130, 260, 70), (9, 2048, 512), (4608, 32, 128), (300, 96, 1000)])
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_layouts(ops, M, N, K, ta, tb):
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=gen)
    Bm = torch.randn((N, K) if tb else (K, N), generator=gen)
    ref = (A.double().T if ta else A.double()) @ (Bm.double().T if tb else Bm.double())
    out = ops.gemm(A.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_NONE)
    assert relerr(out, ref) < 2e-6


def test_gemm_epilogues(ops):
    gen = torch.Generator().manual_seed(3)
    M, N, K = 200, 136, 72
    A, Bm = torch.randn((M, K), generator=gen), torch.randn((N, K), generator=gen)
    bias, R1, R2 = torch.randn(N, generator=gen), torch.randn((M, N), generator=gen), torch.randn((M, N), generator=gen)
    acc = A.double() @ Bm.double().T
    cu = lambda t: t.cuda()
    out = ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_BIAS, bias=cu(bias))
    assert relerr(out, acc + bias.double()) < 2e-6
    d2 = torch.empty((M, N), device="cuda")
    out = ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_BIAS_GELU, bias=cu(bias), D2=d2)
    assert relerr(out, acc + bias.double()) < 2e-6
    assert relerr(d2, F.gelu(acc + bias.double())) < 2e-6
    out = ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_BIAS_RES, bias=cu(bias), R1=cu(R1), R2=cu(R2))
    assert relerr(out, acc + bias.double() + R1.double() + R2.double()) < 2e-6
    out = ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_BIAS_RES, bias=cu(bias), R1=cu(R1))
    assert relerr(out, acc + bias.double() + R1.double()) < 2e-6
    z = R1.double().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(z).sum(), z)
    out = ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_MUL_DGELU, R1=cu(R1))
    assert relerr(out, acc * dg) < 2e-6
    base = cu(R2).clone()
    ops.gemm(cu(A), cu(Bm), M, N, K, 0, 1, ops.EPI_ACCUM, out=base)
    assert relerr(base, acc + R2.double()) < 2e-6


@pytest.mark.parametrize("M,N,K,ta,tb", [(288, 24576, 96, 0, 1), (576, 16384, 72, 0, 0), (288, 22528, 40, 1, 0),
                                         (576, 16400, 64, 1, 1)])
def test_gemm_96_row_tiles(ops, M, N, K, ta, tb):
    """The 96 x 128 tile of the f32 GEMM (one row of four waves, three accumulator tiles each): chosen for the bottleneck
    level's 288- / 576-row matrices when whole 96-row tiles fill the chip; all four layouts, a ragged last column tile,
    and an epilogue with row-indexed inputs (BIAS_RES on the same launch geometry)."""
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=gen)
    Bm = torch.randn((N, K) if tb else (K, N), generator=gen)
    ref = (A.double().T if ta else A.double()) @ (Bm.double().T if tb else Bm.double())
    out = ops.gemm(A.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_NONE)
    assert relerr(out, ref) < 2e-6
    bias, R1 = torch.randn(N, generator=gen), torch.randn((M, N), generator=gen)
    out = ops.gemm(A.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_BIAS_RES, bias=bias.cuda(), R1=R1.cuda())
    assert relerr(out, ref + bias.double() + R1.double()) < 2e-6


def test_gemm_splitk_weight_gradient_shape(ops):
    # dW[128,32] = dY^T X over 73,728 pixel rows: the split-K + atomics path
    gen = torch.Generator().manual_seed(4)
    Mpix, Co, Ci = 73728, 128, 32
    dY, X = torch.randn((Mpix, Co), generator=gen), torch.randn((Mpix, Ci), generator=gen)
    acc = torch.zeros((Co, Ci), device="cuda")
    ops.gemm(dY.cuda(), X.cuda(), Co, Ci, Mpix, 1, 0, ops.EPI_ACCUM, out=acc)
    ref = dY.double().T @ X.double()
    assert relerr(acc, ref) < 5e-6
    acc2 = torch.zeros((Co, Ci), device="cuda")
    ops.gemm(dY.cuda(), X.cuda(), Co, Ci, Mpix, 1, 0, ops.EPI_ACCUM, out=acc2, allow_splitk=False)
    assert relerr(acc2, ref) < 2e-5        # one sequential f32 chain of 73,728 terms per element


# ------------------------------------------------------------------ LayerNorm, dwconv, conv3x3, colsum, adam
@pytest.mark.parametrize("rows,C", [(500, 32), (77, 128), (300, 512), (40, 2048), (9, 8192), (64, 3), (10, 12), (5, 700),
                                    (4099, 8), (1000, 16), (333, 64), (2500, 256), (700, 1024), (130, 4096), (50, 520),
                                    (20000, 32), (600, 2048), (300, 8192), (1000, 3072)])
def test_layernorm_fwd_bwd(ops, rows, C):
    gen = torch.Generator().manual_seed(rows + C)
    x = torch.randn((rows, C), generator=gen) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    gy = torch.randn((rows, C), generator=gen)
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    rgx, rgg, rgb = torch.autograd.grad(ref, [xr, gr, br], gy.double())
    y, mean, rstd = ops.layer_norm(x.cuda(), gamma.cuda(), beta.cuda())
    assert relerr(y, ref) < 5e-6
    gg, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gx = ops.layer_norm_bwd(x.cuda(), gamma.cuda(), mean, rstd, gy.cuda(), gg, gb)
    assert relerr(gx, rgx) < 2e-5
    assert relerr(gg, rgg) < 2e-5 and relerr(gb, rgb) < 2e-5


DW_SHAPES = [(2, 48, 48, 32), (3, 6, 6, 2048), (2, 3, 3, 8192), (1, 5, 7, 8), (2, 12, 12, 128), (1, 20, 33, 12),
             (2, 24, 24, 36), (2, 16, 12, 6), (5, 6, 6, 40), (3, 3, 3, 5), (1, 8, 8, 4), (2, 14, 21, 64)]


@pytest.mark.parametrize("B,H,W,C", DW_SHAPES)
def test_dwconv7_paths_agree(ops, B, H, W, C):
    """LDS-tiled / whole-image kernels against the generic kernel: same accumulation order, so the forward
    and the data gradient (flip + residual) are bit-identical; the weight gradient sums in another order."""
    gen = torch.Generator().manual_seed(7 * B + H + C)
    x, r = torch.randn((B, H, W, C), generator=gen).cuda(), torch.randn((B, H, W, C), generator=gen).cuda()
    w, b = (torch.randn((C, 1, 7, 7), generator=gen) * 0.1).cuda(), torch.randn(C, generator=gen).cuda()
    out = {}
    for seg in (16, 0):                          # 16: generic kernels (explicit per-call choice), 0: chosen by shape
        gw, gb = torch.zeros_like(w), torch.zeros(C, device="cuda")
        ops.dwconv7_weight_grad(x, r, gw, gb, seg=seg)
        ops.dwconv7_weight_grad(x, r, gw, gb, seg=seg)              # accumulates: twice the gradient
        out[seg] = (ops.dwconv7(x, w, b, seg=seg), ops.dwconv7(x, w, None, flip=True, res=r, res_scale=2.0, seg=seg),
                    gw, gb)
        torch.cuda.synchronize()
    assert torch.equal(out[16][0], out[0][0]) and torch.equal(out[16][1], out[0][1])
    assert relerr(out[0][2], out[16][2]) < 2e-6 and relerr(out[0][3], out[16][3]) < 2e-6


PIPE_SHAPES = [(2, 48, 48, 32), (3, 24, 24, 128), (2, 12, 12, 512), (1, 20, 27, 32), (2, 9, 8, 64), (1, 14, 21, 96),
               (5, 16, 16, 32), (1, 96, 96, 32), (2, 48, 32, 128), (3, 8, 8, 32)]


@pytest.mark.parametrize("B,H,W,C", PIPE_SHAPES)
def test_dwconv7_pipelined_kernel_matches_the_first_tiled_kernel(ops, B, H, W, C):
    """The LDS-DMA double-buffered kernel (sei_dwconv7_fwd_ex seg 66: what seg 0 picks for C % 32 == 0) against the
    first tiled kernel (seg 65): same accumulation order, so forward and data gradient (flipped taps + scaled
    residual) are bit-identical; ragged tiles, several tiles / channel groups per workgroup, batch edges."""
    gen = torch.Generator().manual_seed(3 * B + H + C)
    x, r = torch.randn((B, H, W, C), generator=gen).cuda(), torch.randn((B, H, W, C), generator=gen).cuda()
    w, b = (torch.randn((C, 1, 7, 7), generator=gen) * 0.1).cuda(), torch.randn(C, generator=gen).cuda()
    out = {}
    for seg in (65, 66, 0):
        out[seg] = (ops.dwconv7(x, w, b, seg=seg), ops.dwconv7(x, w, None, flip=True, res=r, res_scale=2.0, seg=seg))
        torch.cuda.synchronize()
    for seg in (66, 0):
        assert torch.equal(out[65][0], out[seg][0]) and torch.equal(out[65][1], out[seg][1]), seg
    again = ops.dwconv7(x, w, b, seg=66)
    assert torch.equal(again, out[66][0])


@pytest.mark.parametrize("out16", [False, True])
@pytest.mark.parametrize("B,H,W,C", [(2, 48, 48, 32), (3, 24, 24, 128), (1, 20, 27, 32), (2, 9, 8, 128), (5, 16, 16, 32),
                                     (2, 12, 12, 512), (2, 6, 6, 2048), (1, 14, 21, 64), (1, 96, 96, 32)])
def test_dwconv7_layernorm_fused_forward(ops, B, H, W, C, out16):
    """sei_dwconv7_ln_fwd[_ex] (ConvBlock.conv1 -> LayerNorm, convolutional.py:36-39): ONE launch for C = 32 / 128 (a
    workgroup owns every channel of its pixels), two elsewhere. h1 bit-identical to the depthwise kernel; h2, mean, rstd
    against the stand-alone LayerNorm kernel on that h1 (another summation order: 2e-6) and against float64."""
    import _native
    gen = torch.Generator().manual_seed(B + 2 * H + C)
    x = (torch.randn((B, H, W, C), generator=gen) * 1.5 + 0.3).cuda()
    w, b = (torch.randn((C, 1, 7, 7), generator=gen) * 0.1).cuda(), torch.randn(C, generator=gen).cuda()
    gamma, beta = torch.randn(C, generator=gen).cuda(), torch.randn(C, generator=gen).cuda()
    assert (_native.lib().sei_dwconv7_ln_fwd_launches(B, H, W, C) == 1) == (C == 32 and H >= 8 and W >= 8)
    can_fuse = C in (32, 128) and H >= 8 and W >= 8           # (C = 128: on request, it only ties with the two launches)
    h1, h2, mean, rstd = ops.dwconv7_ln(x, w, b, gamma, beta, out16=out16, fuse=1 if can_fuse else 0)
    if can_fuse:                                              # the two-launch form of the same entry point agrees
        g1, g2, gm, gr = ops.dwconv7_ln(x, w, b, gamma, beta, out16=out16, fuse=2)
        assert torch.equal(g1, h1) and relerr(gm, mean) < 2e-6 and relerr(gr, rstd) < 2e-6
    ref1 = ops.dwconv7(x, w, b, seg=65 if H >= 8 and W >= 8 else 0)
    assert torch.equal(h1, ref1)
    M = B * H * W
    ref64 = F.layer_norm(ref1.double().cpu().view(M, C), (C,), gamma.double().cpu(), beta.double().cpu(), eps=1e-6)
    if out16:
        y, m2, r2 = ops.layer_norm16(ref1.view(M, C), gamma, beta)
        assert h2.dtype == torch.bfloat16
        assert float((h2.double().cpu() - ref64).abs().max()) <= 2 ** -8 * float(ref64.abs().max()) + 1e-6
        # the two kernels round the same f32 value except where their last bits straddle a bf16 tie
        assert float((h2.float() != y.float()).float().mean()) < 2e-3
    else:
        y, m2, r2 = ops.layer_norm(ref1.view(M, C), gamma, beta)
        assert relerr(h2, y) < 2e-6 and relerr(h2, ref64) < 5e-6
    assert relerr(mean, m2) < 2e-6 and relerr(rstd, r2) < 2e-6
    assert relerr(mean, ref1.double().view(M, C).mean(1)) < 1e-5


@pytest.mark.parametrize("kind,B,H,W,C", [("down", 3, 48, 48, 32), ("up", 2, 24, 24, 128), ("down", 2, 24, 24, 128),
                                          ("down", 5, 28, 24, 64), ("up", 1, 24, 24, 48), ("down", 2, 40, 28, 48),
                                          ("down", 300, 24, 24, 16), ("up", 3, 12, 12, 512), ("down", 2, 12, 12, 64),
                                          ("down", 2, 14, 12, 32), ("up", 70, 12, 12, 64)])
def test_resampler_on_the_matrix_cores(ops, kind, B, H, W, C):
    """sei_sepmap2_bf16 (Ideal{Down,Up}sample as two small-GEMM passes on v_mfma_f32_16x16x32_bf16, bf16 mode) against
    the f32 kernel and against float64 on the matrices of models/_mats.py: activations are rounded to bf16 (x, and the
    intermediate between the two products), the matrices are split into head + remainder, so the error is a few bf16
    roundings of the data; forward and transposed (backward) maps, ragged tiles, non-square images."""
    import _native
    from models import _mats
    rate = 2
    fwd, bwd = _mats.resample_matrices(kind, H, W, rate, "cuda")
    gen = torch.Generator().manual_seed(B + H + C)
    x = torch.randn((B, H, W, C), generator=gen).cuda()
    Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
    assert _native.lib().sei_sepmap2_bf16_eligible(B, H, W, Ho, Wo, C) == 1
    for mats, xin, ho, wo in ((fwd, x, Ho, Wo), (bwd, torch.randn((B, Ho, Wo, C), generator=gen).cuda(), H, W)):
        if not _native.lib().sei_sepmap2_bf16_eligible(xin.shape[0], xin.shape[1], xin.shape[2], ho, wo, C):
            continue                                        # (the transposed 96 -> 48 map of the widest case)
        y16 = ops.sepmap2_16(xin, mats, ho, wo)
        y32 = ops.sepmap2(xin, mats, ho, wo)
        L1, R1, L2, R2 = (m.double().cpu() for m in mats[:4])
        xd = xin.double().cpu()
        ref = torch.einsum("pi,bijc,qj->bpqc", L1, xd, R1) + torch.einsum("pi,bijc,qj->bpqc", L2, xd, R2)
        assert relerr(y32, ref) < 5e-6
        scale = float(ref.abs().max())
        err = float((y16.double().cpu() - ref).abs().max()) / scale
        assert err < 1.5e-2, err                            # max-norm: a few 2^-9 roundings of the largest terms
        rms = float((y16.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert rms < 4e-3, rms
        # exact on data that bf16 represents: integers through a 0/1 matrix-free check is not available here, so the
        # run-to-run determinism (no atomics) is asserted instead
        assert torch.equal(y16, ops.sepmap2_16(xin, mats, ho, wo))
        yb = ops.sepmap2_16(xin, mats, ho, wo, out16=True)           # sei_sepmap2_bf16_out16: the same accumulators, rounded once
        assert yb.dtype == torch.bfloat16 and torch.equal(yb, y16.bfloat16())
    assert _native.lib().sei_sepmap2_bf16_eligible(2, 6, 6, 12, 12, 2048) == 0     # small levels stay on the f32 kernels
    assert _native.lib().sei_sepmap2_bf16_eligible(2, 12, 12, 24, 24, 512) == 1    # (round 5: input extents from 12)
    assert _native.lib().sei_sepmap2_bf16_eligible(2, 8, 8, 16, 16, 512) == 0
    assert _native.lib().sei_sepmap2_bf16_eligible(2, 96, 96, 192, 192, 128) == 0  # the x4 network's fine levels: sei_sepmap2_big


@pytest.mark.parametrize("kind,B,H,W,C", [("down", 64, 6, 6, 2048), ("up", 2, 3, 3, 8192), ("up", 5, 6, 6, 2048),
                                          ("down", 2, 8, 6, 64), ("up", 1, 2, 2, 128), ("down", 3, 8, 4, 192),
                                          ("down", 700, 4, 4, 64), ("up", 3, 8, 8, 128), ("down", 40, 12, 12, 1024)])
def test_resampler_of_the_deep_levels_in_one_pass(ops, kind, B, H, W, C):
    """sei_sepmap2_small (Ideal{Down,Up}sample at input extents <= 8: the 6- and 3-pixel images of a 48-pixel crop,
    reference src/models/convolutional.py:54-92,113-133) against float64 on the matrices of models/_mats.py and against the
    two-launch float32 kernels it replaces in the bf16 mode: float32 FMAs, so the float32 bar (5e-6) holds; forward and
    transposed (backward) maps, non-square images, more items than resident workgroups, every register-array width."""
    import _native
    from models import _mats
    fwd, bwd = _mats.resample_matrices(kind, H, W, 2, "cuda")
    gen = torch.Generator().manual_seed(B + H + C)
    x = torch.randn((B, H, W, C), generator=gen).cuda()
    Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
    assert _native.lib().sei_sepmap2_small_eligible(B, H, W, Ho, Wo, C) == 1
    ran = 0
    for mats, xin, ho, wo in ((fwd, x, Ho, Wo), (bwd, torch.randn((B, Ho, Wo, C), generator=gen).cuda(), H, W)):
        if not _native.lib().sei_sepmap2_small_eligible(xin.shape[0], xin.shape[1], xin.shape[2], ho, wo, C):
            continue                                        # (the transposed map of an upsampler to 24 pixels: input 24)
        ran += 1
        _native.record_calls(True)
        y = ops.sepmap2_16(xin, mats, ho, wo)
        assert [n for n, _ in _native.record_calls(False)] == ["sei_sepmap2_small"]
        y32 = ops.sepmap2(xin, mats, ho, wo)
        L1, R1, L2, R2 = (m.double().cpu() for m in mats[:4])
        xd = xin.double().cpu()
        ref = torch.einsum("pi,bijc,qj->bpqc", L1, xd, R1) + torch.einsum("pi,bijc,qj->bpqc", L2, xd, R2)
        assert y.shape == ref.shape
        assert relerr(y, ref) < 5e-6, relerr(y, ref)
        assert relerr(y, y32) < 5e-6
        assert torch.equal(y, ops.sepmap2_16(xin, mats, ho, wo))     # no atomics: run-to-run identical
        y16 = ops.sepmap2_16(xin, mats, ho, wo, out16=True)          # the bf16 result = the float32 one rounded once
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y.bfloat16())
    assert ran >= 1
    assert _native.lib().sei_sepmap2_small_eligible(2, 24, 24, 12, 12, 128) == 0     # sei_sepmap2_bf16's extents
    assert _native.lib().sei_sepmap2_small_eligible(2, 6, 6, 3, 3, 32) == 0          # C % 64
    assert _native.lib().sei_sepmap2_small_eligible(2, 12, 12, 6, 6, 512) == 0       # 12 x 12 inputs, few items: the others win
    assert _native.lib().sei_sepmap2_small_eligible(96, 12, 12, 6, 6, 2048) == 1     # ... reduced to 6 x 6 on >= 512 wave items
    assert _native.lib().sei_sepmap2_small_eligible(96, 12, 12, 24, 24, 512) == 0    # 12 -> 24 stays on the matrix cores


@pytest.mark.parametrize("B,H,W,C", DW_SHAPES[:6])
def test_dwconv7(ops, B, H, W, C):
    from _native import call
    gen = torch.Generator().manual_seed(B + H + C)
    x = torch.randn((B, C, H, W), generator=gen)
    w, b = torch.randn((C, 1, 7, 7), generator=gen) * 0.1, torch.randn(C, generator=gen)
    gy = torch.randn((B, C, H, W), generator=gen)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=3, groups=C)
    rgx, rgw, rgb = torch.autograd.grad(ref, [xr, wr, br], gy.double())
    y = ops.dwconv7(nhwc(x).cuda(), w.cuda(), b.cuda())
    assert relerr(nchw(y), ref) < 5e-6
    gx = ops.dwconv7(nhwc(gy).cuda(), w.cuda(), None, flip=True)
    assert relerr(nchw(gx), rgx) < 5e-6
    gw, gb = torch.zeros_like(w).cuda(), torch.zeros(C, device="cuda")
    xd, gyd = nhwc(x).cuda(), nhwc(gy).cuda()        # keep the buffers alive across the launch
    ops.dwconv7_weight_grad(xd, gyd, gw, gb)
    torch.cuda.synchronize()
    assert relerr(gw, rgw) < 2e-5 and relerr(gb, rgb) < 2e-5


@pytest.mark.parametrize("Ci,Co,nchw_in,nchw_out", [(3, 32, True, False), (32, 3, False, True), (3, 8, False, False),
                                                    (8, 3, False, False)])
def test_conv3x3(ops, Ci, Co, nchw_in, nchw_out):
    gen = torch.Generator().manual_seed(Ci * 10 + Co)
    B, H, W = 2, 20, 24
    x = torch.randn((B, Ci, H, W), generator=gen)
    w, b = torch.randn((Co, Ci, 3, 3), generator=gen) * 0.2, torch.randn(Co, generator=gen)
    res = torch.randn((B, Co, H, W), generator=gen)
    gy = torch.randn((B, Co, H, W), generator=gen)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1) + res.double()
    rgx, rgw, rgb = torch.autograd.grad(ref, [xr, wr, br], gy.double())
    xin = (x if nchw_in else nhwc(x)).cuda().requires_grad_(True)
    wp, bp = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    rin = (res if nchw_out else nhwc(res)).cuda()
    y = ops.Conv3x3Fn.apply(xin, wp, bp, rin, nchw_in, nchw_out)
    assert relerr(y if nchw_out else nchw(y), ref) < 5e-6
    y.backward((gy if nchw_out else nhwc(gy)).cuda())
    assert relerr(xin.grad if nchw_in else nchw(xin.grad), rgx) < 5e-6
    assert relerr(wp.grad, rgw) < 2e-5 and relerr(bp.grad, rgb) < 2e-5


@pytest.mark.parametrize("Ci,Co,nchw_small,B,H,W", [(3, 32, 1, 3, 48, 48), (32, 3, 1, 3, 48, 48), (3, 32, 0, 2, 20, 24),
                                                    (32, 3, 0, 2, 20, 24), (1, 32, 1, 1, 7, 5), (32, 2, 0, 5, 9, 1)])
def test_end_convolution_weight_gradients_on_the_matrix_cores(ops, Ci, Co, nchw_small, B, H, W):
    """conv3x3_wgrad_mfma_kernel (v_mfma_f32_32x32x2_f32: exact float32 products) through both entry points -- sei_conv3x3_bwd_
    weight (atomics; accumulates into gw / gb) and sei_conv3x3_bwd_weight_parts (per-workgroup rows, folded by sei_fold_many)
    -- against float64 torch.autograd of F.conv2d (src/models/convolutional.py:174-176: UNet.in_conv / out_conv): both
    layouts of the small tensor, a single small channel, one-pixel-wide images, ranges that end inside a batch of 16 pixels."""
    import ctypes
    import _native as N
    gen = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn((B, Ci, H, W), generator=gen)
    gy = torch.randn((B, Co, H, W), generator=gen)
    w = torch.zeros((Co, Ci, 3, 3), dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    rgw, rgb = torch.autograd.grad(F.conv2d(x.double(), w, bias, padding=1), [w, bias], gy.double())
    small_is_x = Ci <= 3
    lay = lambda t, small: (t if (small and nchw_small) else nhwc(t)).contiguous().cuda()
    xd, gyd = lay(x, small_is_x), lay(gy, not small_is_x)
    nchw_x, nchw_gy = int(small_is_x and nchw_small), int((not small_is_x) and nchw_small)
    base_w, base_b = torch.randn((Co, Ci, 3, 3), generator=gen).cuda(), torch.randn(Co, generator=gen).cuda()
    gw, gb = base_w.clone(), base_b.clone()
    N.call("sei_conv3x3_bwd_weight", xd.data_ptr(), gyd.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, H, W, Ci, Co, nchw_x, nchw_gy)
    assert relerr(gw - base_w, rgw) < 2e-5 and relerr(gb - base_b, rgb) < 2e-5
    parts = N.lib().sei_conv3x3_bwd_weight_parts_count(B, H, W, Ci, Co, nchw_x, nchw_gy)
    assert parts >= 1
    ncol = Co * Ci * 9 + Co
    work = torch.full((parts, ncol), float("nan"), device="cuda")
    N.call("sei_conv3x3_bwd_weight_parts", xd.data_ptr(), gyd.data_ptr(), work.data_ptr(), B, H, W, Ci, Co, nchw_x, nchw_gy)
    assert torch.isfinite(work).all()                          # every entry of every row is written
    total = work.double().sum(0)
    assert relerr(total[:Co * Ci * 9].view(Co, Ci, 3, 3), rgw) < 2e-5 and relerr(total[Co * Ci * 9:], rgb) < 2e-5
    gw2, gb2 = base_w.clone(), base_b.clone()                  # ... and sei_fold_many adds the rows into the gradients
    job = N.FoldJob()
    job.a, job.b, job.c = gw2.data_ptr(), gb2.data_ptr(), None
    job.ncol, job.split, job.kind, job.nseg = ncol, Co * Ci * 9, N.FOLD_SPLIT, 1
    job.part[0], job.groups[0] = work.data_ptr(), parts
    N.call("sei_fold_many", (N.FoldJob * 1)(job), 1)
    assert relerr(gw2 - base_w, rgw) < 2e-5 and relerr(gb2 - base_b, rgb) < 2e-5
    assert N.lib().sei_conv3x3_bwd_weight_parts_count(B, H, W, 8, 8, 0, 0) == 0     # other shapes: the FMA kernels


def test_colsum_and_adam(ops):
    from _native import call
    gen = torch.Generator().manual_seed(8)
    for M, N in [(1000, 128), (73728, 32), (50, 3), (288, 8192), (17, 300)]:
        X = torch.randn((M, N), generator=gen)
        acc = torch.ones(N, device="cuda")
        ops.colsum_into(acc, X.cuda())
        assert relerr(acc, X.double().sum(0) + 1) < 1e-5
    n = 100003
    p0, g = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.999), foreach=False)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gs = g * step
        ref_p.grad = gs.clone()
        opt.step()
        gd = gs.cuda()
        p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        call("sei_adam_fused", p.data_ptr(), gd.data_ptr(), 0, m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999,
             1e-8, 0.0, step, 1.0, p16.data_ptr())
        torch.cuda.synchronize()
        assert relerr(p, ref_p) < 1e-6
        assert torch.equal(p16, p.bfloat16())
    # bf16-compressed gradients (the multi-GPU exchange format): same update as the f32 kernel fed the rounded grads
    pa, pb = p0.clone().cuda(), p0.clone().cuda()
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    g16 = g.cuda().bfloat16()
    g16f = g16.float()
    call("sei_adam_fused", pa.data_ptr(), g16.data_ptr(), 1, ma.data_ptr(), va.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0,
         1, 0.5, None)
    call("sei_adam_fused", pb.data_ptr(), g16f.data_ptr(), 0, mb.data_ptr(), vb.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.0,
         1, 0.5, None)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb)


# ------------------------------------------------------------------ ideal resamplers
def test_ideal_resamplers_vs_golden(golden):
    from models.convolutional import IdealDownsample, IdealUpsample
    g = golden("g7_ideal_resamplers")
    for H in [48, 24, 12, 6, 20]:
        y = IdealDownsample(2)(nhwc(dev(g[f"ideal_down2.{H}.x"])))
        assert relerr(nchw(y), g[f"ideal_down2.{H}.y64"]) < 5e-6
    for H in [3, 6, 12, 24, 10]:
        y = IdealUpsample(2)(nhwc(dev(g[f"ideal_up2.{H}.x"])))
        assert relerr(nchw(y), g[f"ideal_up2.{H}.y64"]) < 5e-6
    for rate, H in [(4, 12), (4, 48), (3, 18)]:
        y = IdealUpsample(rate)(nhwc(dev(g[f"ideal_up{rate}.{H}.x"])))
        assert relerr(nchw(y), g[f"ideal_up{rate}.{H}.y"]) < 5e-6
    assert relerr(nchw(IdealDownsample(2)(nhwc(dev(g["ideal_down2.rect.x"])))), g["ideal_down2.rect.y"]) < 5e-6
    assert relerr(nchw(IdealUpsample(2)(nhwc(dev(g["ideal_up2.rect.x"])))), g["ideal_up2.rect.y"]) < 5e-6
    with pytest.raises(RuntimeError):      # the reference raises for this shape too
        IdealUpsample(3)(torch.rand(1, 16, 16, 3, device="cuda"))
    # adjointness of the backward (transposed matrices)
    # (seeded inputs, inner products in float64: the two f32 sums of ~18k products otherwise differ by ~1e-5)
    gen = torch.Generator().manual_seed(12)
    x = torch.rand((2, 12, 12, 16), generator=gen).cuda().requires_grad_(True)
    y = IdealUpsample(2)(x)
    z = torch.rand(tuple(y.shape), generator=gen).cuda()
    (gx,) = torch.autograd.grad(y, x, z)
    lhs, rhs = (y.double() * z.double()).sum(), (x.double() * gx.double()).sum()
    assert abs(lhs - rhs) / lhs.abs() < 1e-5


# ------------------------------------------------------------------ layers vs goldens
def _load_module(mod, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}
    mod.load_state_dict(sd)
    return mod.cuda()


def _check_layer(mod, g, name, tol_y=1e-5, tol_g=5e-5):
    x = nhwc(dev(g[f"{name}.x"])).requires_grad_(True)
    y = mod(x)
    assert relerr(nchw(y), g[f"{name}.y"]) < tol_y
    mod.zero_grad()
    y.backward(nhwc(dev(g[f"{name}.ct"])))
    assert relerr(nchw(x.grad), g[f"{name}.gx"]) < tol_g
    for k, p in mod.named_parameters():
        assert relerr(p.grad, g[f"{name}.grad.{k}"]) < tol_g, k


def test_layers_vs_golden(golden):
    from models import convolutional as C
    g = golden("g7_layers")
    _check_layer(_load_module(C.ConvBlock(16), g, "convblock16.sd."), g, "convblock16")
    _check_layer(_load_module(C.ConvBlock(8), g, "convblock8.sd."), g, "convblock8")
    _check_layer(_load_module(C.LayerNorm(12, eps=1e-6), g, "layernorm12.sd."), g, "layernorm12")
    _check_layer(_load_module(C.Downsample(in_channels=8), g, "downsample8.sd."), g, "downsample8")
    _check_layer(_load_module(C.Upsample(in_channels=32, rate=2), g, "upsample32.sd."), g, "upsample32")
    _check_layer(_load_module(C.Upsample(in_channels=3, out_channels=3, rate=4), g, "upsample3x4.sd."), g,
                 "upsample3x4")


# ------------------------------------------------------------------ whole model vs goldens
@pytest.mark.parametrize("tag", ["h8s3_deblur", "h8s3_sr2", "h2s4_pad", "h8s2_nb2", "h2s4_sr4"])
def test_unet_vs_golden(golden, tag):
    from models.convolutional import ConvolutionalModel
    g = golden(f"g7_unet_{tag}")
    cfg = json.loads(bytes(g["cfg"]).decode())
    m = _load_module(ConvolutionalModel(**cfg), g, "sd.")
    assert m.flat_params is not None and m.flat_params.is_cuda
    x = dev(g["x"]).requires_grad_(True)
    y = m(x)
    assert tuple(y.shape) == g["y"].shape
    # Bar: as accurate as the reference's own float32 path. The float64 goldens are the truth; the
    # float32 oracle (same op sequence as the reference, run here on the CPU) measures how far a
    # float32 evaluation of THIS network sits from it -- some of these tiny configs normalise over 2
    # or 3 channels and are ill-conditioned in float32 for the reference too.
    sd32 = {k[3:]: torch.from_numpy(g[k].copy()).requires_grad_(True) for k in g.files if k.startswith("sd.")}
    x32 = torch.from_numpy(g["x"].copy()).requires_grad_(True)
    y32 = tp.unet_forward(sd32, x32, **cfg)
    g32 = torch.autograd.grad(y32, [x32] + list(sd32.values()), torch.from_numpy(g["ct"].copy()))
    base = {"y": relerr(y32, g["y64"]), "gx": relerr(g32[0], g["gx64"])}
    for (k, _), gv in zip(sd32.items(), g32[1:]):
        base[k] = relerr(gv, g[f"grad.{k}"])
    assert relerr(y, g["y64"]) < max(1e-5, 4 * base["y"])
    m.zero_grad()
    y.backward(dev(g["ct"]))
    assert relerr(x.grad, g["gx64"]) < max(1e-4, 8 * base["gx"])
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert relerr(p.grad, g[f"grad.{k}"]) < max(1e-4, 8 * base[k]), (k, relerr(p.grad, g[f"grad.{k}"]), base[k])
    # gradients live in the flat bucket, and accumulate across backward passes
    assert p.grad.data_ptr() >= m.flat_grads.data_ptr()
    before = m.flat_grads.clone()
    m(x.detach()).backward(dev(g["ct"]))
    assert relerr(m.flat_grads, 2 * before) < 1e-5
    m.zero_grad_flat()
    assert m.flat_grads.abs().max() == 0


def test_unet_default_width_vs_oracle():
    """Default hidden_channels=32 but 3 scales (channels 32/128/512) at the training crop size."""
    from models.convolutional import ConvolutionalModel
    torch.manual_seed(0)
    cfg = dict(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True, num_conv_blocks=1,
               hidden_channels=32, inout_convs=True, scales=3)
    m = ConvolutionalModel(**cfg)
    sd = {k: v.clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    x = torch.rand((4, 3, 48, 48), generator=gen)
    ct = torch.randn((4, 3, 48, 48), generator=gen)
    ref = tp.unet_forward(sd, x.double(), **cfg)
    grads = torch.autograd.grad(ref, list(sd.values()), ct.double())
    m = m.cuda()
    y = m(x.cuda())
    assert relerr(y, ref) < 1e-5
    y.backward(ct.cuda())
    got = dict(m.named_parameters())
    for (k, _), gref in zip(sd.items(), grads):
        assert relerr(got[k].grad, gref) < 1e-4, k


def test_model_factory_surface():
    import argparse
    import models
    import physics
    args = argparse.Namespace(
        task="deblurring", kernel="Gaussian_R2", sr_factor=None, noise_level=5, physics_v2=True,
        physics_true_adjoint=False, model_kind="Proposed", ProposedModel__architecture="Convolutional",
        ConvolutionalModel__residual=True, ConvolutionalModel__inner_residual=True,
        ConvolutionalModel__num_conv_blocks=1, ConvolutionalModel__inout_convs=True,
        ConvolutionalModel__hidden_channels=8, ConvolutionalModel__scales=3, data_parallel_devices=None)
    p = physics.get_physics(args, "cuda")
    model = models.get_model(args, p, "cuda")
    model.to("cuda")
    model.train()
    y = torch.rand(2, 3, 48, 48, device="cuda")
    out = model(y, p)                      # extra positional arguments are ignored, as upstream
    assert out.shape == y.shape and torch.equal(out, model(y))
    w = model.get_weights()
    assert "seq.0.in_conv.weight" in w
    model.load_weights({k: v.clone() for k, v in w.items()})
    args.ProposedModel__architecture = "Transformer"               # the reference default: SwinIR (tests/test_swinir_gpu.py)
    assert type(models.get_model(args, p, "cuda").get_backbone()).__name__ == "SwinIR"
    args.data_parallel_devices = "0,1"
    with pytest.raises(NotImplementedError, match="torch.distributed.run --nnodes=1 --nproc-per-node 2"):
        models.get_model(args, p, "cuda")
    args.data_parallel_devices = None
    args.ProposedModel__architecture = "Nope"
    with pytest.raises(ValueError):
        models.get_model(args, p, "cuda")


# ------------------------------------------------------------------ bf16 throughput mode
@pytest.mark.parametrize("M,N,K", [(288, 512, 128), (2304, 128, 32), (37, 29, 19), (130, 260, 70), (576, 2048, 512),
                                    (4608, 32, 128), (300, 96, 1000), (1, 3, 3)])
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_bf16_layouts(ops, M, N, K, ta, tb):
    """bf16 MFMA path: exact for bf16-representable inputs up to f32 accumulation order."""
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=gen).bfloat16().float()
    Bm = torch.randn((N, K) if tb else (K, N), generator=gen).bfloat16().float()
    ref = (A.double().T if ta else A.double()) @ (Bm.double().T if tb else Bm.double())
    prev = ops.set_compute_dtype("bf16")
    try:
        out = ops.gemm(A.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_NONE)
        bias = torch.randn(N, generator=gen)
        d2 = torch.empty((M, N), device="cuda")
        out2 = ops.gemm(A.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_BIAS_GELU, bias=bias.cuda(), D2=d2)
        # un-rounded inputs: the kernel rounds to bf16 (RNE) itself
        A2 = torch.randn((K, M) if ta else (M, K), generator=gen)
        ref2 = (A2.bfloat16().double().T if ta else A2.bfloat16().double()) @ (Bm.double().T if tb else Bm.double())
        out3 = ops.gemm(A2.cuda(), Bm.cuda(), M, N, K, ta, tb, ops.EPI_NONE)
    finally:
        ops.set_compute_dtype(prev)
    assert relerr(out, ref) < 3e-6
    assert relerr(out2, ref + bias.double()) < 3e-6 and relerr(d2, F.gelu(ref + bias.double())) < 3e-6
    assert relerr(out3, ref2) < 3e-6


def test_bf16_mode_model_quality():
    """Same weights, bf16 GEMMs vs f32 GEMMs: restored images agree to bf16 rounding and their PSNR
    against a clean target differs by < 0.01 dB."""
    import metrics
    from models import _ops
    from models.convolutional import ConvolutionalModel
    torch.manual_seed(0)
    m = ConvolutionalModel(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True, num_conv_blocks=1,
                           hidden_channels=32, inout_convs=True, scales=4).cuda()
    gen = torch.Generator().manual_seed(2)
    x = torch.rand((4, 3, 48, 48), generator=gen)
    y = (x + 5 / 255 * torch.randn((4, 3, 48, 48), generator=gen)).cuda()
    with torch.no_grad():
        ref = m(y)
        prev = _ops.set_compute_dtype("bf16")
        try:
            got = m(y)
        finally:
            _ops.set_compute_dtype(prev)
    assert relerr(got, ref) < 2e-2
    for i in range(4):
        assert abs(float(metrics.psnr_fn(got[i].cpu(), x[i])) - float(metrics.psnr_fn(ref[i].cpu(), x[i]))) < 0.01
    # and the backward runs in bf16 mode with gradients close to the f32 ones
    ct = torch.randn(ref.shape, generator=gen).cuda()
    m.zero_grad_flat()
    m(y).backward(ct)
    g32 = m.flat_grads.clone()
    prev = _ops.set_compute_dtype("bf16")
    try:
        m.zero_grad_flat()
        m(y).backward(ct)
    finally:
        _ops.set_compute_dtype(prev)
    cos = torch.nn.functional.cosine_similarity(g32, m.flat_grads, dim=0)
    assert cos > 0.999, float(cos)


@pytest.mark.parametrize("M,N,K", [(576, 2048, 512), (288, 512, 128), (192, 256, 64), (100, 300, 192), (1152, 8192, 2048),
                                    (576, 512, 8192), (2304, 128, 512), (700, 260, 128)])
def test_gemm_bf16nt(ops, M, N, K):
    """Direct-to-LDS bf16 kernel: exact for bf16 inputs up to f32 accumulation order; all epilogues."""
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((M, K), generator=gen).bfloat16()
    Bm = torch.randn((N, K), generator=gen).bfloat16()
    bias, R1 = torch.randn(N, generator=gen), torch.randn((M, N), generator=gen)
    ref = A.double() @ Bm.double().T
    Ad, Bd = A.cuda(), Bm.cuda()
    o32 = torch.empty((M, N), device="cuda")
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_NONE, out32=o32)
    assert relerr(o32, ref) < 3e-6
    o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    g16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_BIAS_GELU, out32=o32, out16=o16, bias=bias.cuda(), D2_16=g16)
    assert relerr(o32, ref + bias.double()) < 3e-6
    assert relerr(o16.float(), (ref + bias.double())) < 5e-3 and relerr(g16.float(), F.gelu(ref + bias.double())) < 5e-3
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_BIAS_RES, out32=o32, bias=bias.cuda(), R1=R1.cuda(), R2=R1.cuda())
    assert relerr(o32, ref + bias.double() + 2 * R1.double()) < 3e-6
    z = R1.double().requires_grad_(True)
    (dg,) = torch.autograd.grad(F.gelu(z).sum(), z)
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_MUL_DGELU, out16=o16, R1=R1.cuda())
    assert relerr(o16.float(), ref * dg) < 5e-3


@pytest.mark.parametrize("M,Np,Kp", [(576, 512, 128), (288, 2048, 512), (9216, 128, 32), (304, 136, 72)])
def test_weight_gradient_reduction_major(ops, M, Np, Kp):
    """dW += dY^T X straight from the (M, .) bf16 tensors (no transposes), bias gradient from the cast pass."""
    gen = torch.Generator().manual_seed(M + Np)
    dY, X = torch.randn((M, Np), generator=gen), torch.randn((M, Kp), generator=gen).bfloat16()
    base = torch.randn((Np, Kp), generator=gen)
    ref = base.double() + dY.bfloat16().double().T @ X.double()
    acc = base.clone().cuda()
    cs = torch.ones(Np, device="cuda")
    dY16 = ops.cast16(dY.cuda(), colsum_into_=cs)
    assert relerr(cs, dY.double().sum(0) + 1) < 1e-5
    assert torch.equal(dY16.cpu(), dY.bfloat16())
    ops.weight_grad16(dY16, X.cuda(), acc)
    assert relerr(acc, ref) < 5e-6


@pytest.mark.parametrize("M,N,K", [(2304, 512, 128), (576, 2048, 512), (288, 512, 2048), (72, 8192, 2048), (4608, 128, 512)])
def test_split_bf16_gemm_against_float64(ops, M, N, K):
    """models/_ops.py gemm_x3 (--compute_dtype bf16x3): every orientation and epilogue the float32 layer functions use,
    against float64 -- a float32-class result (relative error of a few 1e-6 of the largest entry; one bf16 product alone is
    ~3e-3) from three bf16 MFMA launches; and the head / remainder planes themselves."""
    gen = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=gen)
    planes = ops.split_x2(x.cuda())
    hi, lo = planes[0].float().cpu(), planes[1].float().cpu()
    assert torch.equal(hi, x.bfloat16().float()) and torch.equal(lo, (x - hi).bfloat16().float())
    assert float((x - hi - lo).abs().max() / x.abs().max()) < 2.0 ** -16
    prev = ops.set_compute_dtype("bf16x3")
    try:
        w = 0.05 * torch.randn((N, K), generator=gen)
        bias, res = torch.randn((N,), generator=gen), torch.randn((M, N), generator=gen)
        xd, wd, bd, rd = x.cuda(), w.cuda(), bias.cuda(), res.cuda()
        ref = x.double() @ w.double().T
        scale = float(ref.abs().max())
        tol = 1.2e-5                  # (operands carry 2^-18 each, the dropped a_lo b_lo another 2^-18; measured up to 6e-6)
        # forward orientation: A (M, K), B (N, K); bias + GELU with the second output
        h4 = torch.empty((M, N), device="cuda")
        h3 = ops.gemm(xd, wd, M, N, K, 0, 1, ops.EPI_BIAS_GELU, bias=bd, D2=h4)
        want = ref + bias.double()
        assert float((h3.cpu().double() - want).abs().max()) < tol * scale
        assert float((h4.cpu().double() - torch.nn.functional.gelu(want)).abs().max()) < tol * scale
        out = ops.gemm(xd, wd, M, N, K, 0, 1, ops.EPI_BIAS_RES, bias=bd, R1=rd, R2=rd)
        assert float((out.cpu().double() - (want + 2 * res.double())).abs().max()) < tol * scale
        s = torch.rand((M,), generator=gen)
        out = ops.gemm(xd, wd, M, N, K, 0, 1, ops.EPI_BIAS_ROWSCALE, bias=bd, R1=s.cuda())
        assert float((out.cpu().double() - (ref + s.double()[:, None] * bias.double())).abs().max()) < tol * scale
        # data gradient: B (K', N') read reduction-major; plain and with GELU'
        g = torch.randn((M, N), generator=gen)
        gd = g.cuda()
        gref = g.double() @ w.double()
        gx = ops.gemm(gd, wd, M, K, N, 0, 0, ops.EPI_NONE)
        assert float((gx.cpu().double() - gref).abs().max()) < tol * float(gref.abs().max())
        pre = torch.randn((M, K), generator=gen)
        gx = ops.gemm(gd, wd, M, K, N, 0, 0, ops.EPI_MUL_DGELU, R1=pre.cuda())
        p64 = pre.double().requires_grad_(True)
        (dg,) = torch.autograd.grad(torch.nn.functional.gelu(p64).sum(), p64)
        assert float((gx.cpu().double() - gref * dg).abs().max()) < tol * float(gref.abs().max())
        # weight gradient: both operands reduction-major, accumulated into a running gradient
        base = torch.randn((N, K), generator=gen)
        acc = base.clone().cuda()
        ops.gemm(gd, xd, N, K, M, 1, 0, ops.EPI_ACCUM, out=acc)
        wref = g.double().T @ x.double()
        assert float((acc.cpu().double() - (wref + base.double())).abs().max()) < tol * float(wref.abs().max())
    finally:
        ops.set_compute_dtype(prev)


@pytest.mark.parametrize("same_stream", [False, True])
def test_splitk_gemm_inside_hipgraph(ops, same_stream):
    """A split-K GEMM replayed from a captured graph must not accumulate stale sums. same_stream: warm-up and capture on
    one side stream, so the launch meets its (device, stream) slab workspace (what graphs.GraphedLossStep does); otherwise
    the capturing stream has no workspace yet and the launch takes the zero-fill + float-atomics path."""
    gen = torch.Generator().manual_seed(5)
    M, N, K = 2304, 128, 512                      # 18 tiles, long K: the split-K path
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    Bm = torch.randn((N, K), generator=gen).bfloat16().cuda()
    ref = A.double().cpu() @ Bm.double().cpu().T
    out = torch.full((M, N), 7.0, device="cuda")
    side = torch.cuda.Stream()
    ops._SPLITK_WS.pop((0, side.cuda_stream), None)       # (torch hands out pooled streams: forget an earlier user's)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side if same_stream else torch.cuda.current_stream()):
        ops.gemm_nt16(A, Bm, M, N, K, ops.EPI_NONE, out32=out)          # warm up outside capture
    torch.cuda.synchronize()
    assert ((0, side.cuda_stream) in ops._SPLITK_WS) == same_stream
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        assert (ops.splitk_workspace("cuda:0")[0] is not None) == same_stream
        ops.gemm_nt16(A, Bm, M, N, K, ops.EPI_NONE, out32=out)
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert relerr(out, ref) < 3e-6


def test_splitk_workspaces_belong_to_one_stream_each(ops):
    """Two split-K launches in flight on two streams use two workspaces (the ticket protocol indexes counters and slabs
    by tile ordinal alone); both results are right, and reset_splitk_counters leaves every counter block zero."""
    gen = torch.Generator().manual_seed(6)
    M, N, K = 2304, 128, 2048
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    B1, B2 = (torch.randn((N, K), generator=gen).bfloat16().cuda() for _ in range(2))
    outs = [torch.empty((M, N), device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for _ in range(20):
        for st, Bm, out in zip(streams, (B1, B2), outs):
            with torch.cuda.stream(st):
                ops.gemm_nt16(A, Bm, M, N, K, ops.EPI_NONE, out32=out)
    torch.cuda.synchronize()
    mine = [(0, st.cuda_stream) for st in streams]
    assert all(k in ops._SPLITK_WS for k in mine) and ops._SPLITK_WS[mine[0]] is not ops._SPLITK_WS[mine[1]]
    assert sum(1 for k in ops._SPLITK_WS if k[0] == 0) <= ops.SPLITK_WS_STREAMS      # least recently used ones are dropped
    for Bm, out in zip((B1, B2), outs):
        assert relerr(out, A.double().cpu() @ Bm.double().cpu().T) < 3e-6
    ops.reset_splitk_counters("cuda:0")
    torch.cuda.synchronize()
    for key in mine:
        assert int(ops._SPLITK_WS[key][:ops.SPLITK_COUNTER_BYTES].max()) == 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (576, 512, 128), (288, 2048, 512), (136, 264, 72), (2304, 128, 512),
                                    (512, 128, 9216), (8, 16, 24)])
@pytest.mark.parametrize("arm,brm", [(False, True), (True, True), (True, False)])
def test_gemm_bf16nt_reduction_major_operands(ops, M, N, K, arm, brm):
    """Operands stored reduction-major ((K,M) / (K,N)) are consumed through transposing LDS reads."""
    gen = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((M, K), generator=gen).bfloat16()
    Bm = torch.randn((N, K), generator=gen).bfloat16()
    ref = A.double() @ Bm.double().T
    Ad = (A.t().contiguous() if arm else A).cuda()
    Bd = (Bm.t().contiguous() if brm else Bm).cuda()
    out = torch.empty((M, N), device="cuda")
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_NONE, out32=out, a_rmajor=arm, b_rmajor=brm)
    assert relerr(out, ref) < 3e-6
    base = torch.randn((M, N), generator=gen)
    acc = base.clone().cuda()
    ops.gemm_nt16(Ad, Bd, M, N, K, ops.EPI_ACCUM, out32=acc, a_rmajor=arm, b_rmajor=brm)
    assert relerr(acc, ref + base.double()) < 3e-6


@pytest.mark.parametrize("M,N,K1,K2", [(128, 32, 147456, 73728), (512, 2048, 576, 288), (2048, 512, 2304, 1152),
                                       (96, 160, 1000, 24), (32, 128, 40, 8),
                                       (2048, 8192, 288, 576), (8192, 2048, 1152, 2304), (6144, 2560, 24, 296)])
def test_gemm_bf16nt_two_segment_weight_gradient(ops, M, N, K1, K2):
    """sei_gemm_bf16nt_dw2: D (+)= A1^T B1 + A2^T B2 on reduction-major operands, store and accumulate."""
    from _native import call
    gen = torch.Generator().manual_seed(M + N + K1)
    A1, A2 = torch.randn((K1, M), generator=gen).bfloat16().cuda(), torch.randn((K2, M), generator=gen).bfloat16().cuda()
    B1, B2 = torch.randn((K1, N), generator=gen).bfloat16().cuda(), torch.randn((K2, N), generator=gen).bfloat16().cuda()
    ref = A1.double().T @ B1.double() + A2.double().T @ B2.double()
    D = torch.full((M, N), float("nan"), device="cuda")                # store must not read D
    call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N, D.data_ptr(), M, N,
         K1, K2, 0)
    assert relerr(D, ref) < 2e-5
    call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), N, D.data_ptr(), M, N,
         K1, K2, 1)
    assert relerr(D, 2 * ref) < 2e-5


@pytest.mark.parametrize("br", [0, 1])
@pytest.mark.parametrize("code", [30, 31, 32, 33])
@pytest.mark.parametrize("M,N,K", [(576, 512, 256), (300, 520, 328), (2304, 2048, 2048), (288, 1024, 4096), (1000, 128, 32)])
def test_gemm_bf16_quadrant_schedule(ops, M, N, K, code, br):
    """The quadrant schedule (gemm_bf16pq.h; tile codes 30-33 = 256/288 rows x 256/128 columns): ragged row and
    column edges, a K that is not a multiple of the k-tile (also a single partial k-tile), split K, every fused epilogue, against float64; launches that do not split K must agree bit
    for bit run to run (the schedule orders its LDS-DMA by counted waits and raw barriers only)."""
    gen = torch.Generator().manual_seed(M + N + K + code + br)
    kw = dict(b_rmajor=bool(br), tile=code)
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    B = torch.randn((K, N) if br else (N, K), generator=gen).bfloat16().cuda()
    bias, R1 = torch.randn(N, generator=gen).cuda(), torch.randn((M, N), generator=gen).cuda()
    R2, rs = torch.randn((M, N), generator=gen).cuda(), torch.randn(M, generator=gen).cuda()
    ref = (A.double() @ (B.double() if br else B.double().t())).cpu()
    outs = []
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out16=out, **kw)            # bf16 output: never split
        torch.cuda.synchronize()
        outs.append(out)
    assert relerr(outs[0].float(), ref) < 1e-2
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out32=out, **kw)                 # f32 output: may split K
    assert relerr(out, ref) < 2e-5
    out = torch.empty((M, N), device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1, R2=R2, **kw)
    assert relerr(out, ref + bias.double().cpu() + R1.double().cpu() + R2.double().cpu()) < 2e-5
    h3, h4 = torch.empty((M, N), device="cuda"), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_GELU, out32=h3, bias=bias, D2_16=h4, **kw)
    pre = ref + bias.double().cpu()
    assert relerr(h3, pre) < 2e-5 and relerr(h4.float(), F.gelu(pre)) < 1e-2
    g16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_MUL_DGELU, out16=g16, R1=R1, **kw)
    x = R1.double().cpu().requires_grad_(True)
    dg, = torch.autograd.grad(F.gelu(x).sum(), x)
    assert relerr(g16.float(), ref * dg) < 1e-2
    acc = R1.clone()
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_ACCUM, out32=acc, **kw)
    assert relerr(acc, ref + R1.double().cpu()) < 2e-5
    out = torch.empty((M, N), device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_ROWSCALE, out32=out, bias=bias, R1=rs, **kw)
    assert relerr(out, ref + rs.double().cpu()[:, None] * bias.double().cpu()[None, :]) < 2e-5


@pytest.mark.parametrize("M,N,K,kind", [
    (2304, 8192, 2048, "gelu"), (576, 32768, 8192, "gelu"), (288, 32768, 8192, "gelu"), (36864, 512, 128, "gelu"),
    (147456, 128, 32, "gelu"), (2304, 8192, 2048, "dgelu_kr"), (576, 32768, 8192, "dgelu_kr"),
    (147456, 128, 32, "dgelu_kr"), (2304, 2048, 8192, "res"), (576, 8192, 32768, "res"), (576, 8192, 32768, "none_kr"),
    (2304, 8192, 2048, "none_kr")])
def test_gemm_bf16_full_size_layers_on_the_automatic_dispatch(ops, M, N, K, kind):
    """The 1x1-convolution GEMMs of BASELINE configs[1] at their full sizes (batch 32), as the library dispatches
    them (quadrant schedule for most, split K where it splits): f32 results against a float32 matmul of the same
    bf16 operands, bf16 results to bf16 resolution; no run-to-run difference where K is not split."""
    gen = torch.Generator().manual_seed(M + N + K)
    kr = kind.endswith("_kr")
    A = (torch.randn((M, K), generator=gen) * 0.5).bfloat16().cuda()
    B = (torch.randn((K, N) if kr else (N, K), generator=gen) * (K ** -0.5)).bfloat16().cuda()
    bias = torch.randn(N, generator=gen).cuda()
    ref = A.float() @ (B.float() if kr else B.float().t())
    scale = float(ref.abs().max())
    if kind == "gelu":
        outs = []
        for _ in range(2):
            h3 = torch.empty((M, N), device="cuda")
            h4 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_GELU, out32=h3, bias=bias, D2_16=h4)
            outs.append((h3, h4))
        pre = ref + bias
        assert float((outs[0][0] - pre).abs().max()) < 2e-5 * max(scale, 1.0) * (K ** 0.5)
        assert float((outs[0][1].float() - F.gelu(pre)).abs().max()) < 1e-2 * max(scale, 1.0)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    elif kind == "dgelu_kr":
        R1 = torch.randn((M, N), generator=gen).cuda()
        x = R1.clone().requires_grad_(True)
        dg, = torch.autograd.grad(F.gelu(x).sum(), x)
        outs = []
        for _ in range(2):
            g16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            ops.gemm_nt16(A, B, M, N, K, ops.EPI_MUL_DGELU, out16=g16, R1=R1, b_rmajor=True)
            outs.append(g16)
        assert float((outs[0].float() - ref * dg).abs().max()) < 1e-2 * max(scale, 1.0)
        assert torch.equal(outs[0], outs[1])
    elif kind == "res":
        R1 = torch.randn((M, N), generator=gen).cuda()
        out = torch.empty((M, N), device="cuda")
        ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_RES, out32=out, bias=bias, R1=R1)
        assert float((out - (ref + bias + R1)).abs().max()) < 2e-5 * max(scale, 1.0) * (K ** 0.5)
    else:
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out32=out, b_rmajor=True)
        assert float((out - ref).abs().max()) < 2e-5 * max(scale, 1.0) * (K ** 0.5)


@pytest.mark.parametrize("code", [30, 33])
@pytest.mark.parametrize("M,N,K", [(512, 768, 256), (304, 520, 328), (2048, 2048, 864), (256, 1024, 4104)])
def test_gemm_bf16_quadrant_schedule_weight_gradient(ops, M, N, K, code):
    """Quadrant schedule with both operands reduction-major (the weight gradient): ragged edges, a K that is not
    a multiple of the k-tile (zero rows past the end), store and accumulate, bit-stable without split K."""
    gen = torch.Generator().manual_seed(M + N + K + code)
    A = torch.randn((K, M), generator=gen).bfloat16().cuda()
    B = torch.randn((K, N), generator=gen).bfloat16().cuda()
    R1 = torch.randn((M, N), generator=gen).cuda()
    ref = (A.double().t() @ B.double()).cpu()
    outs = []
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out16=out, a_rmajor=True, b_rmajor=True, tile=code)
        torch.cuda.synchronize()
        outs.append(out)
    assert relerr(outs[0].float(), ref) < 1e-2
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out32=out, a_rmajor=True, b_rmajor=True, tile=code)
    assert relerr(out, ref) < 2e-5
    acc = R1.clone()
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_ACCUM, out32=acc, a_rmajor=True, b_rmajor=True, tile=code)
    assert relerr(acc, ref + R1.double().cpu()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(2304, 8192, 2048), (300, 520, 72)])
def test_gemm_bf16nt_bias_rowscale_and_weighted_colsum(ops, M, N, K):
    """D = A B^T + bias[n] * s[m] (the bias of a 1x1 convolution that was moved behind the ideal downsampler)
    and its gradient, the row-weighted column sum."""
    from _native import call
    gen = torch.Generator().manual_seed(M + K)
    A, B = torch.randn((M, K), generator=gen).bfloat16().cuda(), torch.randn((N, K), generator=gen).bfloat16().cuda()
    bias, s = torch.randn(N, generator=gen).cuda(), torch.randn(M, generator=gen).cuda()
    out = torch.empty((M, N), device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_BIAS_ROWSCALE, out32=out, bias=bias, R1=s)
    ref = A.double() @ B.double().t() + s.double()[:, None] * bias.double()[None, :]
    assert relerr(out, ref) < 2e-5
    go = torch.randn((M, N), generator=gen).cuda()
    gb = torch.ones(N, device="cuda")
    ops.colsum_into(gb, go, row_weight=s)
    assert relerr(gb, 1.0 + (go.double() * s.double()[:, None]).sum(0)) < 2e-5


@pytest.mark.parametrize("M,N", [(147456, 128), (9216, 2048), (576, 32768), (1000, 24), (333, 50), (77, 7)])
def test_bf16_cast_and_column_sums(ops, M, N):
    """sei_colsum_bf16 (16-byte lanes / scalar fallback) and the streaming cast + column-sum pass."""
    gen = torch.Generator().manual_seed(M + N)
    X = torch.randn((M, N), generator=gen)
    X16 = X.bfloat16().cuda()
    acc = torch.full((N,), 2.0, device="cuda")
    ops.colsum16_into(acc, X16)
    assert relerr(acc, 2.0 + X16.double().sum(0)) < 2e-6
    cs = torch.full((N,), -1.0, device="cuda")
    Y16 = ops.cast16(X.cuda(), colsum_into_=cs)
    assert torch.equal(Y16.cpu(), X.bfloat16())
    assert relerr(cs, -1.0 + X.double().sum(0)) < 2e-6
    assert torch.equal(ops.cast16(X.cuda()).cpu(), X.bfloat16())


@pytest.mark.parametrize("rows,C", [(40, 2048), (9, 8192), (130, 4096), (701, 1024), (50, 520), (5, 700), (300, 512),
                                    (500, 32), (2304, 2048), (577, 8192)])
def test_layernorm_fwd_bf16_output(ops, rows, C):
    """sei_ln_fwd_bf16 (group / 16-byte-lane / legacy wide kernels): bf16 output within one rounding of the
    float64 LayerNorm, float32 statistics as the f32 kernel's."""
    gen = torch.Generator().manual_seed(rows * 3 + C)
    x = torch.randn((rows, C), generator=gen) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    y16, mean, rstd = ops.layer_norm16(x.cuda(), gamma.cuda(), beta.cuda())
    ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), eps=1e-6)
    assert y16.dtype == torch.bfloat16
    assert float((y16.double().cpu() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-6
    assert relerr(mean, x.double().mean(1)) < 1e-5
    assert relerr(rstd, 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-6)) < 1e-5


@pytest.mark.parametrize("kind,B,H,W,C", [("down", 2, 192, 192, 32), ("up", 2, 96, 96, 128), ("down", 3, 96, 96, 128),
                                          ("down", 1, 256, 256, 32), ("up", 1, 128, 128, 128), ("down", 2, 96, 128, 48),
                                          ("up", 70, 96, 96, 16), ("up", 2, 48, 48, 128), ("up", 2, 64, 64, 32)])
def test_resampler_on_the_matrix_cores_at_large_extents(ops, kind, B, H, W, C):
    """sei_sepmap2_big (csrc/sepmap_big.hip: both products of Ideal{Down,Up}sample as one batched constant-matrix GEMM
    kernel, bf16 intermediate; the 96- / 192-pixel levels of the x4 network, the 256-pixel inputs of the un-cropped
    series; reference src/models/convolutional.py:54-133) against the f32 kernel and float64 on the matrices of
    models/_mats.py, forward and transposed maps: the same bars as the small-extent kernel (a few bf16 roundings of the
    data; the matrices are head + remainder). Non-square images, a channel count that leaves a 48-wide last tile, more items
    than workgroups; bitwise repeatable (no atomics)."""
    import _native
    from models import _mats
    fwd, bwd = _mats.resample_matrices(kind, H, W, 2, "cuda")
    gen = torch.Generator().manual_seed(B + H + C)
    x = torch.randn((B, H, W, C), generator=gen).cuda()
    Ho, Wo = fwd[0].shape[0], fwd[1].shape[0]
    ran = 0
    for mats, xin, ho, wo in ((fwd, x, Ho, Wo), (bwd, torch.randn((B, Ho, Wo, C), generator=gen).cuda(), H, W)):
        if not _native.lib().sei_sepmap2_big_eligible(xin.shape[0], xin.shape[1], xin.shape[2], ho, wo, C):
            continue                                        # (a transposed map whose input is small enough for sei_sepmap2_bf16)
        ran += 1
        y16 = ops.sepmap2_16(xin, mats, ho, wo)
        y32 = ops.sepmap2(xin, mats, ho, wo)
        L1, R1, L2, R2 = (m.double().cpu() for m in mats[:4])
        xd = xin.double().cpu()
        ref = torch.einsum("pi,bijc,qj->bpqc", L1, xd, R1) + torch.einsum("pi,bijc,qj->bpqc", L2, xd, R2)
        assert relerr(y32, ref) < 5e-6
        scale = float(ref.abs().max())
        err = float((y16.double().cpu() - ref).abs().max()) / scale
        assert err < 1.5e-2, err
        rms = float((y16.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert rms < 4e-3, rms
        assert torch.equal(y16, ops.sepmap2_16(xin, mats, ho, wo))
    assert ran >= 1
    assert _native.lib().sei_sepmap2_big_eligible(2, 48, 48, 24, 24, 128) == 0      # small extents: sei_sepmap2_bf16's
    assert _native.lib().sei_sepmap2_big_eligible(2, 96, 96, 48, 48, 3) == 0        # the x4 pre-upsampler's 3 channels


@pytest.mark.parametrize("twice", [False, True])
@pytest.mark.parametrize("B,H,W,C", [(2, 5, 7, 32), (3, 48, 48, 32), (1, 9, 11, 128), (2, 24, 24, 128), (1, 12, 12, 128),
                                     (16, 24, 24, 128), (1, 128, 128, 128), (2, 128, 128, 32)])
def test_fused_mlp_block_matches_the_unfused_bf16_block(ops, B, H, W, C, twice):
    """The fused pointwise MLP of the shallow levels (sei_mlp_fused_fwd / _bwd: conv2 -> GELU -> conv3 + residual
    with the hidden activation in registers, recomputed by the backward) against the unfused bf16 block on the same
    weights: identical bf16 products and roundings, so outputs and gradients agree to accumulation-order noise; and
    against a float64 evaluation of the block to bf16 resolution. Ragged pixel counts (not multiples of 32); at C = 128
    pixel counts that are multiples of 144 take the nine-wave kernel of csrc/mlp128.hip (one workgroup, eight, 64); from 9216
    pixels on, counts that are not (the 128 x 128 grids: 16,384 and 32,768 pixels) are split into whole 144-pixel groups on
    that kernel and a tail on the first fused kernel; small counts that are not: the GEMMs (sei_mlp_fused_eligible)."""
    prev = ops.set_compute_dtype("bf16")
    saved = ops.FUSED_MLP_CHANNELS
    try:
        gen = torch.Generator().manual_seed(B + H + C + int(twice))
        x = torch.randn((B, H, W, C), generator=gen)
        go = torch.randn((B, H, W, C), generator=gen)

        def params():
            g2 = torch.Generator().manual_seed(99)
            mk = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(s, generator=g2) * sc).cuda())
            return (mk(C, 1, 7, 7, sc=0.1), mk(C, sc=0.1), mk(C, sc=0.3) , mk(C, sc=0.1), mk(4 * C, C, 1, 1, sc=C ** -0.5),
                    mk(4 * C, sc=0.1), mk(C, 4 * C, 1, 1, sc=(4 * C) ** -0.5), mk(C, sc=0.1))

        res = {}
        for mode, chans in (("fused", (32, 128)), ("plain", ())):
            ops.FUSED_MLP_CHANNELS = chans
            ops.weights_updated()
            ps = params()
            with torch.no_grad():
                ps[2].add_(1.0)                                  # LayerNorm weight around 1
            xc = x.cuda().requires_grad_(True)
            ops.begin_step()
            y = ops.ConvBlockFn16.apply(xc, *ps, twice)
            y.backward(go.cuda())
            ops.flush_weight_grads()
            torch.cuda.synchronize()
            res[mode] = (y.detach(), xc.grad, [p.grad.clone() for p in ps])
        assert relerr(res["fused"][0], res["plain"][0]) < 1e-5
        assert relerr(res["fused"][1], res["plain"][1]) < 2e-3
        for a, b in zip(res["fused"][2], res["plain"][2]):
            # (1-D entries: bias gradients. Where the streamed launch serves the weight, conv3's bias gradient is summed from
            # the bf16 copy of go -- the operand of its weight gradient -- instead of the float32 go: ~2^-8 per term)
            assert relerr(a, b) < (5e-3 if a.dim() == 1 else 2e-3)
        # float64 evaluation of the same block
        ps = [p.detach().double().cpu().requires_grad_(True) for p in params()]
        with torch.no_grad():
            ps[2].add_(1.0)
        xd = x.double().requires_grad_(True)
        xn = xd.permute(0, 3, 1, 2)
        h = F.conv2d(xn, ps[0], ps[1], padding=3, groups=C)
        h = F.layer_norm(h.permute(0, 2, 3, 1), (C,), ps[2], ps[3], 1e-6).permute(0, 3, 1, 2)
        h = F.conv2d(F.gelu(F.conv2d(h, ps[4], ps[5])), ps[6], ps[7])
        ref = ((2.0 if twice else 1.0) * xn + h).permute(0, 2, 3, 1)
        ref.backward(go.double())
        assert relerr(res["fused"][0], ref) < 2e-2
        assert relerr(res["fused"][1], xd.grad) < 3e-2
    finally:
        ops.FUSED_MLP_CHANNELS = saved
        ops.set_compute_dtype(prev)


def test_weight_gradient_gemm_with_the_adam_epilogue():
    """sei_gemm_bf16nt_dw2_adam == sei_gemm_bf16nt_dw2 (store) followed by sei_adam_fused over the same (M, N) range:
    parameters, moments and the bf16 shadow bit for bit where the storing GEMM does not split K (768 whole tiles, one-
    stage and two-stage loops), to float32 rounding of the gradient on a ragged shape that it does split;
    sei_adam_scalars hands the epilogue the scalars sei_adam_fused derives itself."""
    import ctypes
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(3)
    for (M, Nn, K1, K2, exact) in ((2048, 6144, 96, 200, True), (2048, 6144, 640, 1032, True), (136, 520, 96, 200, False)):
        A1, A2 = ((0.05 * torch.randn((k, M), device="cuda", generator=gen)).bfloat16() for k in (K1, K2))
        B1, B2 = (torch.randn((k, Nn), device="cuda", generator=gen).bfloat16() for k in (K1, K2))
        p0 = 0.02 * torch.randn((M, Nn), device="cuda", generator=gen)
        host = (ctypes.c_float * 6)()
        N.call("sei_adam_scalars", 2e-4, 0.9, 0.99, 1e-8, 0.01, 5, ctypes.cast(host, ctypes.c_void_p))
        hyper = torch.tensor(list(host), device="cuda")
        state = lambda: (p0.clone(), torch.full_like(p0, 1e-3), torch.full_like(p0, 1e-5),
                         torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16))
        pa, ma, va, sa = state()
        grad = torch.empty((M, Nn), device="cuda")
        N.call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, grad.data_ptr(),
               M, Nn, K1, K2, 0)
        N.call("sei_adam_fused", pa.data_ptr(), grad.data_ptr(), 0, ma.data_ptr(), va.data_ptr(), M * Nn, 2e-4, 0.9, 0.99,
               1e-8, 0.01, 5, 1.0, sa.data_ptr())
        pb, mb, vb, sb = state()
        N.call("sei_gemm_bf16nt_dw2_adam", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn,
               pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), sb.data_ptr(), hyper.data_ptr(), M, Nn, K1, K2)
        assert float((pa - p0).abs().max()) > 1e-5
        for a, b in ((pa, pb), (ma, mb), (va, vb), (sa, sb)):
            assert torch.equal(a, b) if exact else relerr(a.float(), b.float()) < 1e-2 and relerr(ma, mb) < 1e-5
        pc, mc, vc, _ = state()                                   # without a shadow
        N.call("sei_gemm_bf16nt_dw2_adam", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn,
               pc.data_ptr(), mc.data_ptr(), vc.data_ptr(), None, hyper.data_ptr(), M, Nn, K1, K2)
        assert torch.equal(pc, pb)


@pytest.mark.parametrize("M,N,K,brm", [(576, 1024, 512, 1), (288, 2048, 256, 1), (2304, 2048, 512, 0), (300, 520, 328, 1),
                                       (1152, 8192, 2048, 1), (64, 128, 64, 0)])
def test_gemm_with_the_result_s_column_sums_in_the_epilogue(ops, M, N, K, brm):
    """sei_gemm_bf16nt_colsum: D16 = (A op(B)) gelu'(R1) in bf16 and colsum[n] += sum_m D16[m][n] -- against the plain launch
    followed by sei_colsum_bf16 over its result: D16 bit-identical (the same kernel), the sums equal up to the float atomics'
    order. Quadrant tiles of both widths (the sums ride in the epilogue), a ragged shape and a small one (the 128 x 128 loop:
    the column-sum kernel follows), both weight orientations; the sums ACCUMULATE."""
    gen = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    B = (torch.randn((K, N) if brm else (N, K), generator=gen) / K ** 0.5).bfloat16().cuda()
    R1 = torch.randn((M, N), generator=gen).cuda()
    ref16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_MUL_DGELU, out16=ref16, R1=R1, b_rmajor=bool(brm))
    base = torch.randn(N, generator=gen).cuda()
    want = base.clone()
    ops.colsum16_into(want, ref16, leaf=False)
    got16 = torch.empty_like(ref16)
    got = base.clone()
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_MUL_DGELU, out16=got16, R1=R1, b_rmajor=bool(brm), colsum=got)
    assert torch.equal(got16, ref16)
    exact = ref16.double().sum(0) + base.double()
    assert relerr(got, exact) < 2e-6 and relerr(want, exact) < 2e-6
    plain16, plain = torch.empty_like(ref16), torch.zeros(N, device="cuda")
    ops.gemm_nt16(A, B, M, N, K, ops.EPI_NONE, out16=plain16, b_rmajor=bool(brm), colsum=plain)     # the plain product
    assert relerr(plain, plain16.double().sum(0)) < 2e-6


@pytest.mark.parametrize("M,Nn,K,brm,kind,tile", [(1152, 2048, 8192, 0, "res", 32), (576, 2048, 4096, 1, "none", 31),
                                                  (288, 4096, 2048, 0, "gelu", 32), (576, 1024, 2048, 1, "dgelu", 31),
                                                  (512, 640, 1024, 0, "res", 33), (300, 520, 1032, 1, "none", 30),
                                                  (864, 2048, 8192, 1, "none", 0)])
def test_split_k_through_slabs(ops, M, Nn, K, brm, kind, tile):
    """sei_gemm_bf16nt_ws (ABI 11): the quadrant kernel's K slices meet in slabs of a caller-owned workspace -- stored
    write-through, one ticket per slice, the last arriver adds the others and runs the WHOLE epilogue -- against the unsplit
    launch of the same tile: equal to float32 rounding of the regrouped sum for every epilogue (bias + residual, plain,
    bias + GELU with two results, GELU' with a bf16 result and riding column sums), both weight orientations, 288- and
    256-row tiles of both widths, ragged edges, 2 ... 8 slices. Two slices are bit-reproducible (a + b = b + a whichever
    arrives last). The tile counters are zero again after every launch, so launches repeat on the same workspace; the
    slab region is filled with NaNs and READ (the XCDs' L2s then hold stale lines of it) before the launches. tile = 0: the
    automatic choice (cost model of pq_choose_slabs) with and without the workspace."""
    import _native as N
    gen = torch.Generator().manual_seed(M + Nn + K)
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    B = (torch.randn((K, Nn) if brm else (Nn, K), generator=gen) / K ** 0.5).bfloat16().cuda()
    bias, R1 = torch.randn(Nn, generator=gen).cuda(), torch.randn((M, Nn), generator=gen).cuda()
    ws_bytes = 64 << 20
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
    ws[16384:].view(torch.float32).fill_(float("nan"))
    assert bool(torch.isnan(ws[16384:].view(torch.float32).sum()))            # (a pass of plain loads over every slab line)
    epi = {"res": ops.EPI_BIAS_RES, "none": ops.EPI_NONE, "gelu": ops.EPI_BIAS_GELU, "dgelu": ops.EPI_MUL_DGELU}[kind]

    def run(workspace, splitk):
        d32 = torch.full((M, Nn), 7.0, device="cuda") if kind != "dgelu" else None
        d16 = torch.full((M, Nn), 7.0, device="cuda", dtype=torch.bfloat16) if kind == "dgelu" else None
        d2 = torch.full((M, Nn), 7.0, device="cuda", dtype=torch.bfloat16) if kind == "gelu" else None
        cs = torch.ones(Nn, device="cuda") if kind == "dgelu" else None
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), Nn if brm else K, brm, N.ptr(d32), N.ptr(d16), M, Nn, K,
               epi, bias.data_ptr() if kind in ("res", "gelu") else None, R1.data_ptr() if kind in ("res", "dgelu") else None,
               None, N.ptr(d2), N.ptr(cs), None if workspace is None else workspace.data_ptr(),
               0 if workspace is None else ws_bytes, tile, 0, splitk)
        torch.cuda.synchronize()
        return tuple(t for t in (d32, d16, d2, cs) if t is not None)

    def counters():
        return int(ws[:16384].view(torch.int32).abs().sum())

    base = run(None, 1 if tile else 0)
    if tile == 0:
        fam, bm, bn, sk, slabs = N.gemm_plan(0, brm, True, False, M, Nn, K, epi, ws_bytes=ws_bytes)
        assert (fam, bm, bn, slabs) == ("pq", 288, 128, True) and sk > 1, (fam, bm, bn, sk, slabs)
        got = run(ws, 0)
        assert relerr(got[0], base[0]) < 3e-6 and counters() == 0
        return
    tried = 0
    for sk in (2, 3, 4, 8):
        if K // 64 < 4 * sk:
            continue
        tried += 1
        got, again = run(ws, sk), run(ws, sk)
        assert counters() == 0, "tile counters are zero between launches"
        for g_, a_, b_ in zip(got, again, base):
            # (bf16 results: single values move by one ulp where the regrouped float32 sum rounds the other way, and the
            # column sums of those values with them)
            tol = 2e-2 if g_.dtype == torch.bfloat16 else (5e-3 if g_.dim() == 1 else 3e-6)
            assert relerr(g_, b_) < tol and relerr(a_, b_) < tol, (sk, g_.dtype, g_.dim())
            if sk == 2 and g_.dim() == 2:
                assert torch.equal(g_, a_), "two slices: the sum does not depend on which one arrives last"
    assert tried >= 2


def test_split_k_slabs_under_uneven_load(ops):
    """The slab hand-off under the conditions in which a missing release / acquire shows (cdna_hip_programming.md, Guideline
    16, Pitfall 3): 150 launches of two-slice and four-slice GEMMs while a second stream keeps HBM and the L2s busy with large
    copies, so that slices of a tile finish far apart and on busy caches. Two slices must reproduce the first launch's bits
    every time; four slices stay within float32 rounding of it; the counters are zero at the end."""
    import _native as N
    gen = torch.Generator().manual_seed(11)
    M, Nn, K = 1152, 2048, 4096
    A = torch.randn((M, K), generator=gen).bfloat16().cuda()
    B = (torch.randn((Nn, K), generator=gen) / K ** 0.5).bfloat16().cuda()
    bias, R1 = torch.randn(Nn, generator=gen).cuda(), torch.randn((M, Nn), generator=gen).cuda()
    ws_bytes = 64 << 20
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty((M, Nn), device="cuda")
    big_a, big_b = torch.randn(64 << 20, device="cuda"), torch.empty(64 << 20, device="cuda")
    side = torch.cuda.Stream()

    def run(sk):
        N.call("sei_gemm_bf16nt_ws", A.data_ptr(), K, 0, B.data_ptr(), K, 0, out.data_ptr(), None, M, Nn, K, ops.EPI_BIAS_RES,
               bias.data_ptr(), R1.data_ptr(), None, None, None, ws.data_ptr(), ws_bytes, 32, 0, sk)

    run(2)
    torch.cuda.synchronize()
    first = out.clone()
    run(1)
    unsplit = out.clone()
    assert relerr(first, unsplit) < 3e-6
    bad2 = torch.zeros((), device="cuda")
    worst4 = torch.zeros((), device="cuda")
    for it in range(150):
        with torch.cuda.stream(side):
            big_b.copy_(big_a)                               # 512 MB of traffic beside every launch
        run(2)
        bad2 += (out != first).any()
        run(4)
        worst4 = torch.maximum(worst4, (out - first).abs().max())
    side.synchronize()
    torch.cuda.synchronize()
    assert float(bad2) == 0.0, "a two-slice launch changed bits under load"
    assert float(worst4) < 3e-6 * float(first.abs().max())
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M,Nn,K1,K2", [(2048, 6144, 640, 1032), (512, 768, 96, 200), (2048, 8192, 1152, 2304)])
def test_adam_epilogue_of_the_quadrant_schedule(M, Nn, K1, K2):
    """sei_gemm_bf16nt_dw2_adam_ex: the quadrant kernel's Adam epilogue (tiles 30 / 33: 256 x 256 / 256 x 128) against the
    128 x 128 loop's (tile 1), same state, scalars and operands: the two kernels add the same products in another order
    (16x16x32 against 32x32x16 MFMAs), so gradient-dependent results agree to float32 rounding of the sum; the bf16 shadow
    is the rounded new parameter in each; K tails and the straddled K segment included."""
    import ctypes
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(M + K1)
    A1, A2 = ((0.05 * torch.randn((k, M), device="cuda", generator=gen)).bfloat16() for k in (K1, K2))
    B1, B2 = (torch.randn((k, Nn), device="cuda", generator=gen).bfloat16() for k in (K1, K2))
    p0 = 0.02 * torch.randn((M, Nn), device="cuda", generator=gen)
    host = (ctypes.c_float * 6)()
    N.call("sei_adam_scalars", 2e-4, 0.9, 0.99, 1e-8, 0.01, 5, ctypes.cast(host, ctypes.c_void_p))
    hyper = torch.tensor(list(host), device="cuda")
    out = {}
    for tile in (1, 30, 33, 0):
        p, m, v = p0.clone(), torch.full_like(p0, 1e-3), torch.full_like(p0, 1e-5)
        sh = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16)
        N.call("sei_gemm_bf16nt_dw2_adam_ex", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, p.data_ptr(),
               m.data_ptr(), v.data_ptr(), sh.data_ptr(), hyper.data_ptr(), M, Nn, K1, K2, tile)
        out[tile] = (p, m, v, sh)
    assert float((out[1][0] - p0).abs().max()) > 1e-5
    for tile in (30, 33):
        for a, b in zip(out[tile][:3], out[1][:3]):
            assert relerr(a, b) < 2e-5, (tile, relerr(a, b))
        assert torch.equal(out[tile][3], out[tile][0].bfloat16())
    for a, b in zip(out[0], out[1]):                    # the dispatcher keeps these launches on the loop (tools/exp_dw_adam_pq.py)
        assert torch.equal(a, b)


def test_weight_gradient_gemm_with_bf16_output():
    """sei_gemm_bf16nt_dw2_bf16out == the bf16 rounding of the float32 gradient sei_gemm_bf16nt_dw2 stores (no K split
    at 768 whole tiles; ragged tiles to rounding): what the reducer's cast pass would have put into the exchange buffer."""
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(4)
    for (M, Nn, K1, K2, exact) in ((2048, 6144, 96, 200, True), (2048, 6144, 640, 1032, True), (136, 520, 96, 200, False)):
        A1, A2 = ((0.05 * torch.randn((k, M), device="cuda", generator=gen)).bfloat16() for k in (K1, K2))
        B1, B2 = (torch.randn((k, Nn), device="cuda", generator=gen).bfloat16() for k in (K1, K2))
        grad = torch.empty((M, Nn), device="cuda")
        N.call("sei_gemm_bf16nt_dw2", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn, grad.data_ptr(),
               M, Nn, K1, K2, 0)
        out = torch.full((M, Nn), float("nan"), device="cuda", dtype=torch.bfloat16)
        N.call("sei_gemm_bf16nt_dw2_bf16out", A1.data_ptr(), A2.data_ptr(), M, B1.data_ptr(), B2.data_ptr(), Nn,
               out.data_ptr(), M, Nn, K1, K2)
        if exact:
            assert torch.equal(out, grad.bfloat16())
        else:
            assert relerr(out.float(), grad) < 1e-2 and bool(torch.isfinite(out.float()).all())


# ------------------------------------------------------------------ deferred folds (sei_fold_many)
def test_fold_many_matches_the_separate_folds(ops):
    """One sei_fold_many launch against the fold each reducing kernel runs on its own: LayerNorm partial sums (narrow
    and wide shapes), depthwise weight-gradient partial sums, a three-way split with a dropped third output, and a
    destination fed by two launches (the two model calls of a step): bit-identical."""
    import _native as N
    torch.manual_seed(3)
    jobs, keep, expect = [], [], []

    def ln_case(rows_list, C):
        gamma = torch.randn(C, device="cuda")
        gg_ref, gb_ref = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        gg, gb = gg_ref.clone(), gb_ref.clone()
        segs = []
        for rows in rows_list:
            x = torch.randn((rows, C), device="cuda"); gy = torch.randn((rows, C), device="cuda")
            mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
            need = N.lib().sei_ln_bwd_workspace(rows, C)
            parts, off = N.lib().sei_ln_bwd_part_count(rows, C), N.lib().sei_ln_bwd_part_offset(rows, C)
            assert parts > 0
            gx_ref, gx = torch.empty_like(x), torch.empty_like(x)
            w_ref = torch.empty(need, device="cuda"); w = torch.empty(need, device="cuda")
            N.call("sei_ln_bwd", x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(),
                   gx_ref.data_ptr(), gg_ref.data_ptr(), gb_ref.data_ptr(), rows, C, w_ref.data_ptr(), need)
            N.call("sei_ln_bwd", x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gy.data_ptr(),
                   gx.data_ptr(), None, None, rows, C, w.data_ptr(), need)
            assert torch.equal(gx, gx_ref)
            segs.append((w, w.data_ptr() + 4 * off, parts))
        jobs.append((gg, gb, None, 2 * C, C, N.FOLD_SPLIT, segs))
        expect.extend([(gg, gg_ref), (gb, gb_ref)])

    def dw_case(shapes, C):
        gw_ref, gbias_ref = torch.randn((C, 49), device="cuda"), torch.randn(C, device="cuda")
        gw, gbias = gw_ref.clone(), gbias_ref.clone()
        segs = []
        for (B, H, W) in shapes:
            x = torch.randn((B, H, W, C), device="cuda"); gy = torch.randn((B, H, W, C), device="cuda")
            need = N.lib().sei_dwconv7_bwd_weight_workspace(B, H, W, C)
            w_ref = torch.empty(need, device="cuda"); w = torch.empty(need, device="cuda")
            N.call("sei_dwconv7_bwd_weight", x.data_ptr(), gy.data_ptr(), gw_ref.data_ptr(), gbias_ref.data_ptr(), B, H, W, C,
                   w_ref.data_ptr(), need)
            N.call("sei_dwconv7_bwd_weight", x.data_ptr(), gy.data_ptr(), None, None, B, H, W, C, w.data_ptr(), need)
            segs.append((w, w.data_ptr(), need // (50 * C)))
        jobs.append((gw, gbias, None, 50 * C, C, N.FOLD_DWCONV7, segs))
        expect.extend([(gw, gw_ref), (gbias, gbias_ref)])

    ln_case([500], 32); ln_case([1152, 2304], 2048); ln_case([300, 77, 129], 512); ln_case([9], 8192)
    dw_case([(2, 48, 48), (4, 48, 48)], 32); dw_case([(3, 6, 6)], 512); dw_case([(2, 3, 3), (1, 3, 3)], 8192)
    # three sums per group, the third one dropped / kept
    for keep_c in (False, True):
        C, groups = 180, 256
        part = torch.randn((groups, 3, C), device="cuda")
        outs = [torch.randn(C, device="cuda") for _ in range(3)]
        refs = [o + part[:, k].double().sum(0).float() for k, o in enumerate(outs)]
        jobs.append((outs[0], outs[1], outs[2] if keep_c else None, 3 * C, C, N.FOLD_SPLIT, [(part, part.data_ptr(), groups)]))
        keep.append((outs, refs, keep_c))
    arr = (N.FoldJob * len(jobs))()
    for j, (a, b, c, ncol, split, kind, segs) in zip(arr, jobs):
        j.a, j.b, j.c, j.ncol, j.split, j.kind, j.nseg = a.data_ptr(), N.ptr(b), N.ptr(c), ncol, split, kind, len(segs)
        for k, (_, ptr, groups) in enumerate(segs):
            j.part[k] = ptr
            j.groups[k] = groups
    untouched = keep[0][0][2].clone()
    N.call("sei_fold_many", arr, len(jobs))
    torch.cuda.synchronize()
    for got, ref in expect:
        assert torch.equal(got, ref)
    for outs, refs, keep_c in keep:
        for k in range(3 if keep_c else 2):
            assert relerr(outs[k], refs[k]) < 1e-5
    assert torch.equal(keep[0][0][2], untouched)
    # two jobs with one destination are refused (two workgroups would add to the same address)
    dup = (N.FoldJob * 2)()
    dup[0] = arr[0]; dup[1] = arr[0]
    with pytest.raises(N.NativeLibraryError):
        N.call("sei_fold_many", dup, 2)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_deferred_folds_leave_the_gradients_bit_identical(ops, mode):
    """A backward pass of the U-Net with the folds of its reducing kernels deferred to one launch at its end against the
    fold per launch; two model calls per step, so shared parameters are fed by two launches. The folds themselves are
    bit-identical (the test above); whole passes differ by the float atomics of the split-K GEMMs upstream, run to run
    as much as mode to mode, hence a tolerance here."""
    import _native as N
    from models.convolutional import ConvolutionalModel
    torch.manual_seed(0)
    m = ConvolutionalModel(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True, num_conv_blocks=1,
                           hidden_channels=32, inout_convs=True, scales=3).cuda()
    y = torch.rand((2, 3, 24, 24), device="cuda"); y2 = torch.rand((1, 3, 24, 24), device="cuda")
    ct = torch.randn((2, 3, 24, 24), device="cuda"); ct2 = torch.randn((1, 3, 24, 24), device="cuda")
    prev = ops.set_compute_dtype(mode)
    grads, folds = {}, {}
    try:
        for deferred in (False, True):
            ops.DEFERRED_FOLDS = deferred
            ops.set_weight_grad_merging(False, owner=m)      # (merged launches change the summation order of nothing here,
            m.zero_grad_flat()                               #  but keep the two passes independent of the parking logic)
            N.record_calls(True)
            ((m(y) * ct).sum() + (m(y2) * ct2).sum()).backward()
            log = N.record_calls(False)
            folds[deferred] = sum(1 for name, _ in log if name == "sei_fold_many")
            grads[deferred] = m.flat_grads.clone()
    finally:
        ops.DEFERRED_FOLDS = True
        ops.set_weight_grad_merging(True, owner=m)
        ops.set_compute_dtype(prev)
    assert folds[False] == 0 and 1 <= folds[True] <= 2
    assert relerr(grads[True], grads[False]) < (1e-5 if mode == "f32" else 2e-3)


@pytest.mark.parametrize("K1,K2", [(128, 0), (128, 256), (18432, 36864), (147456, 73728)])
def test_streamed_weight_gradients_of_the_shallow_levels(K1, K2):
    """sei_dwstream_bf16_jobs (autograd's weight gradient gy^T x of the 1x1 convolutions of reference
    src/models/convolutional.py:40-42,106,143 at the 32- and 128-channel levels; both operands pixel-major as stored, the
    step's two model calls as two segments) against the float32 product of the same bf16 operands, on top of a running
    gradient (and, where asked, the bias gradient = gy's column sums from the same launch): every block shape the kernel builds -- (128, 512), (512, 128), (128, 256), (32, 128), (128, 32): narrow
    operand first or second, 256-column blocks, pixel pairs read as one row -- each as a table of its own and all of them in
    ONE table; from a single k-tile (most workgroups idle) to more k-tiles than workgroups. The tiled GEMM it replaces
    agrees to the float summation order; shapes the kernel does not build are refused."""
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(K1 + K2)
    jobs, expect = [], []
    for Mo, Ni in [(128, 512), (512, 128), (128, 256), (1024, 128), (32, 128), (128, 32)]:
        gys = [(0.5 * torch.randn((k, Mo), device="cuda", generator=gen)).bfloat16() for k in (K1, K2) if k]
        xs = [torch.randn((k, Ni), device="cuda", generator=gen).bfloat16() for k in (K1, K2) if k]
        base = torch.randn((Mo, Ni), device="cuda", generator=gen)
        ref = base + sum(g.float().T @ x.float() for g, x in zip(gys, xs))
        assert N.lib().sei_dwstream_bf16_eligible(Mo, Ni, Mo, Ni, K1, K2) != 0
        bbase = torch.randn(Mo, device="cuda", generator=gen)
        bref = bbase + sum(g.float().sum(0) for g in gys)                 # the bias gradient: gy's column sums
        job = lambda d, bg=None: N.DwStreamJob(gys[0].data_ptr(), gys[-1].data_ptr(), xs[0].data_ptr(), xs[-1].data_ptr(),
                                               Mo, Ni, Mo, Ni, d.data_ptr(), Ni, 0, K1, K2, N.ptr(bg))
        d, bg = base.clone(), bbase.clone()
        N.call("sei_dwstream_bf16_jobs", (N.DwStreamJob * 1)(job(d, bg)), 1)
        assert relerr(d, ref) < 2e-5, (Mo, Ni, relerr(d, ref))
        assert relerr(bg, bref) < 2e-5, (Mo, Ni, relerr(bg, bref))
        if K2 and (K1 + K2) % 8 == 0:                       # the launch it replaces
            t = base.clone()
            N.call("sei_gemm_bf16nt_dw2", gys[0].data_ptr(), gys[1].data_ptr(), Mo, xs[0].data_ptr(), xs[1].data_ptr(), Ni,
                   t.data_ptr(), Mo, Ni, K1, K2, 1)
            assert relerr(t, d) < 2e-5, (Mo, Ni, relerr(t, d))
        dg = base.clone()
        jobs.append(job(dg))
        expect.append((dg, ref, gys, xs))
    N.call("sei_dwstream_bf16_jobs", (N.DwStreamJob * len(jobs))(*jobs), len(jobs))
    for dg, ref, _, _ in expect:
        assert relerr(dg, ref) < 2e-5, relerr(dg, ref)
    elig = N.lib().sei_dwstream_bf16_eligible
    assert elig(128, 512, 128, 512, 96, 0) == 0 and elig(32, 128, 32, 128, 64, 0) == 0      # ragged pixel counts
    assert elig(128, 512, 136, 512, 128, 0) == 0                                           # padded rows
    assert elig(512, 2048, 512, 2048, 128, 0) == 0 and elig(64, 256, 64, 256, 128, 0) == 0 and elig(128, 128, 128, 128, 128, 0) == 0


def test_streamed_weight_gradients_inside_a_backward_pass(ops):
    """The U-Net's backward pass in bf16 mode with the shallow levels' weight gradients collected into one job table per
    block shape at the end of the pass (models/_ops._queue_dwstream) against the same pass on the tiled GEMMs
    (SEI_NO_DWSTREAM): two model calls per step, so every job carries two pixel segments. Equal up to the float atomics'
    summation order and the bf16 roundings behind it: the bar is three times what two runs of the tiled path differ by
    (or 3e-3); the streamed path issues one launch where the tiled path issues one per weight."""
    import _native as N
    from models.convolutional import ConvolutionalModel
    torch.manual_seed(0)
    m = ConvolutionalModel(in_channels=3, upsampling_rate=1, residual=True, inner_residual=True, num_conv_blocks=1,
                           hidden_channels=32, inout_convs=True, scales=3).cuda()
    y = torch.rand((4, 3, 48, 48), device="cuda"); y2 = torch.rand((2, 3, 48, 48), device="cuda")
    ct = torch.randn((4, 3, 48, 48), device="cuda"); ct2 = torch.randn((2, 3, 48, 48), device="cuda")
    prev = ops.set_compute_dtype("bf16")
    grads, launches = {}, {}
    try:
        for streamed in (False, None, True):                  # (None: the tiled path a second time -- its own noise)
            ops.DWSTREAM = bool(streamed)
            m.zero_grad_flat()
            N.record_calls(True)
            ((m(y) * ct).sum() + (m(y2) * ct2).sum()).backward()
            log = N.record_calls(False)
            launches[streamed] = (sum(1 for name, _ in log if name == "sei_dwstream_bf16_jobs"),
                                  sum(1 for name, _ in log if name.startswith("sei_gemm_bf16nt_dw2")))
            grads[streamed] = m.flat_grads.clone()
    finally:
        ops.DWSTREAM = True
        ops.set_compute_dtype(prev)
    assert launches[False][0] == 0 and launches[True][0] == 1
    assert launches[True][1] <= launches[False][1] - 8, launches          # levels 0 and 1: 2 x (conv2, conv3) + 4 between levels
    # (whole passes differ by the float atomics and the bf16 roundings behind them upstream, run to run as much as path to
    # path: a tolerance, on the gradients the two paths compute differently)
    worst = []
    for name, prm in m.named_parameters():
        if prm.dim() == 4 and prm.shape[-1] == 1 and prm.shape[-2] == 1:
            off = (prm._sei_grad_view.data_ptr() - m.flat_grads.data_ptr()) // 4
            a, b, c = (grads[k][off:off + prm.numel()] for k in (True, False, None))
            worst.append((relerr(a, b) / max(relerr(c, b), 1e-3), relerr(a, b), relerr(c, b), name))
    assert len(worst) == 14 and max(worst)[0] < 3, sorted(worst, reverse=True)[:4]
    # in the streamed launch the bias gradients of the fused-MLP levels are the column sums of the bf16-ROUNDED
    # gradient rows (one more MFMA against a fragment of ones), on the tiled path they are float32 column sums of the
    # un-rounded rows: the same quantity up to the bf16 rounding of each addend (2^-9 relative, signs random: the sums agree
    # far inside that). Held to 3e-3 of the gradient's largest entry.
    checked = 0
    for name, prm in m.named_parameters():
        if name.endswith(("conv2.bias", "conv3.bias")):
            off = (prm._sei_grad_view.data_ptr() - m.flat_grads.data_ptr()) // 4
            a, b = (grads[k][off:off + prm.numel()] for k in (True, False))
            assert relerr(a, b) < 3e-3, (name, relerr(a, b))
            checked += 1
    assert checked >= 10


def test_batched_transposes_of_the_fused_levels_weights(ops):
    """sei_transpose_bf16_many: several (R, C) bf16 matrices -> their (C, R) transposes in one launch, ragged extents
    included; and models/_ops._transposed16_cached rebuilds every remembered stale weight with ONE launch."""
    import _native as N
    gen = torch.Generator().manual_seed(4)
    mats = [torch.randn((r, c), generator=gen).bfloat16().cuda() for r, c in ((128, 32), (32, 128), (512, 128), (128, 512),
                                                                             (70, 200), (1, 5), (64, 64))]
    outs = [torch.full((m.shape[1], m.shape[0]), 7.0, dtype=torch.bfloat16, device="cuda") for m in mats]
    jobs = (N.TransposeJob * len(mats))(*[N.TransposeJob(m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1])
                                         for m, o in zip(mats, outs)])
    N.call("sei_transpose_bf16_many", jobs, len(mats))
    for m, o in zip(mats, outs):
        assert torch.equal(o, m.t().contiguous())
    assert N.lib().sei_transpose_bf16_many(jobs, 0, None) == 10001 and N.lib().sei_transpose_bf16_many(jobs, 17, None) == 10001


@pytest.mark.parametrize("R,C", [(2304, 2048), (300, 512), (147456, 128), (7, 8)])
def test_cast_with_a_weighted_column_sum(ops, R, C):
    """sei_cast_bf16_colsum_weighted: the bf16 copy equals the plain cast's, colsum[c] += sum_r w[r] x[r][c] against float64
    (the Downsample's bias gradient: models/_ops.DownsampleFn16.backward)."""
    gen = torch.Generator().manual_seed(R + C)
    x = torch.randn((R, C), generator=gen).cuda()
    w = torch.rand(R, generator=gen).cuda()
    base = torch.randn(C, generator=gen).cuda()
    acc = base.clone()
    x16 = ops.cast16(x, colsum_into_=acc, row_weight=w)
    assert torch.equal(x16, ops.cast16(x))
    ref = base.double() + (w.double()[:, None] * x.double()).sum(0)
    assert relerr(acc, ref) < 5e-6
