"""GPU parity of the loss layer and of whole training steps against the oracle and the goldens."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import torch_path as tp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel(a, b):
    a, b = float(a), float(b)
    return abs(a - b) / max(abs(b), 1e-30)


def relerr(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def ref_args(**over):
    import bench
    a = bench.reference_args("cuda", hidden=8, scales=3)
    for k, v in over.items():
        setattr(a, k, v)
    return a


# ------------------------------------------------------------------ SURE vs golden G6
@pytest.mark.parametrize("margin", [0, 6])
@pytest.mark.parametrize("cst", [False, True])
def test_sure_vs_golden(golden, margin, cst):
    import physics
    from losses.sure import SureGaussianLoss, embed_probe
    g = golden("g6_sure")
    op = physics.BlurV2(kernel=physics.get_kernel("Gaussian_R2")[None, None])
    y = dev(g["y"])
    w = torch.tensor(0.9, device="cuda", requires_grad=True)
    model = lambda v: w * v + 0.05 * v * v
    lf = SureGaussianLoss(sigma=5 / 255, margin=margin, cropped_div=True, averaged_cst=cst)
    val = lf(y=y, x_net=model(y), physics=op, model=model, b=embed_probe(y, dev(g[f"m{margin}.b"]), margin))
    (gw,) = torch.autograd.grad(val, w)
    assert rel(val, g[f"m{margin}.cst{int(cst)}.loss"]) < 1e-5
    assert rel(gw, g[f"m{margin}.cst{int(cst)}.gw"]) < 1e-4


def test_sure_uncropped_divergence_and_random_probe():
    import physics
    from losses.sure import SureGaussianLoss, draw_probe
    op = physics.BlurV2(kernel=physics.get_kernel("Gaussian_R2")[None, None])
    y = torch.rand(3, 3, 48, 48, device="cuda")
    b = draw_probe(y, 6)
    assert b[:, :, :6].abs().max() == 0 and b[:, :, :, -6:].abs().max() == 0 and abs(float(b[:, :, 6:-6, 6:-6].std()) - 1) < 0.05
    model = lambda v: 0.8 * v
    k = tp.blur_kernel("Gaussian_R2")
    for cropped in (True, False):
        lf = SureGaussianLoss(sigma=5 / 255, margin=6, cropped_div=cropped)
        bb = draw_probe(y, lf.div_margin)
        val = lf(y=y, x_net=model(y), physics=op, model=model, b=bb)
        m = 6 if cropped else 0
        b_int = bb[:, :, m:48 - m, m:48 - m].cpu() if m else bb.cpu()
        ref = tp.sure_loss(y.cpu(), model(y.cpu()), lambda v: tp.blur_fft(v, k), model, 5 / 255, margin=6,
                           cropped_div=cropped, b=b_int)
        assert rel(val, ref) < 2e-5


# ------------------------------------------------------------------ composite proposed loss vs golden G11
def _proposed_manual(model, physics, y, b, noise, rate, center, margin):
    """The product's loss layer driven with injected randomness (b, noise, rate, centre)."""
    import transforms
    from losses.ei import mse
    from losses.sure import SureGaussianLoss
    x_net = model(y)
    sure = SureGaussianLoss(sigma=5 / 255, margin=margin, cropped_div=True, averaged_cst=None)
    l_sure = sure(y=y, x_net=x_net, physics=physics, model=model, b=b)
    with torch.no_grad():
        x2 = transforms.padded_downsampling_transform(x_net.contiguous(), rate, center, "bicubic", "reflection", False)
    y2 = physics.noise_model(physics.A(x2), noise=noise)
    x3 = model(y2)
    l_ei = mse()(x3, x2)
    return l_sure + l_ei, dict(x_net=x_net, x2=x2, x3=x3, loss_sure=l_sure, loss_ei=l_ei)


@pytest.mark.parametrize("tag", ["deblur", "sr2"])
def test_proposed_loss_vs_golden(golden, tag):
    import physics
    from losses.sure import embed_probe
    from models.convolutional import ConvolutionalModel
    g = golden(f"g11_proposed_{tag}")
    up, margin = (1, 6) if tag == "deblur" else (2, 0)
    m = ConvolutionalModel(in_channels=3, upsampling_rate=up, residual=True, inner_residual=True,
                           num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    if tag == "deblur":
        op = physics.BlurV2(kernel=physics.get_kernel("Gaussian_R2")[None, None])
    else:
        op = physics.Downsampling(rate=2, antialias=True)
    op.noise_model = physics.GaussianNoise(sigma=5 / 255)
    y = dev(g["y"])
    loss, aux = _proposed_manual(m, op, y, embed_probe(y, dev(g["b"]), margin), dev(g["n"]), dev(g["rate"]),
                                 dev(g["center"]), margin)
    assert relerr(aux["x_net"], g["x_net"]) < 2e-5
    assert relerr(aux["x2"], g["x2"]) < 2e-5
    assert relerr(aux["x3"], g["x3"]) < 5e-5
    assert rel(aux["loss_sure"], g["loss_sure"]) < 1e-4
    assert rel(aux["loss_ei"], g["loss_ei"]) < 1e-4
    assert rel(loss, g["loss"]) < 1e-4
    m.zero_grad_flat()
    loss.backward()
    for k, p in m.named_parameters():
        assert relerr(p.grad, g[f"grad.{k}"]) < 5e-4, (k, relerr(p.grad, g[f"grad.{k}"]))


# ------------------------------------------------------------------ the reference's in-tree R2R / EI loss, golden G13
@pytest.mark.parametrize("tag", ["deblur", "sr2"])
def test_r2r_and_ei_glue_vs_reference_golden(golden, tag):
    """G13 = src/losses/r2r.py run by the reference's own code on its own U-Net, physics and ScalingTransform
    (tools/gen_golden.py gen_r2r). (i) losses.r2r.R2REILoss on the HIP path: loss terms and every weight gradient;
    (ii) the hot path's losses.ei.EILoss (restated from deepinv) against the same EI term: the reference's ei_loss
    is EILoss with x_net = model(y + 0.5 sigma n1) and measurement noise 1.5 sigma."""
    import physics
    import transforms
    from losses.ei import EILoss, mse
    from losses.r2r import R2REILoss
    from models.convolutional import ConvolutionalModel
    g = golden(f"g13_r2r_{tag}")
    up = 1 if tag == "deblur" else 2
    m = ConvolutionalModel(in_channels=3, upsampling_rate=up, residual=True, inner_residual=True,
                           num_conv_blocks=1, hidden_channels=8, inout_convs=True, scales=3)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    net = lambda v, *ignored: m(v)
    sigma = 5 / 255
    if tag == "deblur":
        op = physics.BlurV2(kernel=physics.get_kernel("Gaussian_R2")[None, None])
    else:
        op = physics.Downsampling(rate=2, antialias=True)
    y, n0, n1, n2 = (dev(g[k]) for k in ("y", "n0", "n1", "n2"))
    # the float64 run's draws, rounded to float32 (rates are exact; centres move by 1e-8)
    params = (dev(g["f64.rate"].astype(np.float32)), dev(g["f64.center"].astype(np.float32)).view(-1, 1, 1, 2))
    scaling = transforms.ScalingTransform(kind="padded", antialias=False)
    T = lambda v: scaling(v, params=params)
    for through in (False, True):
        p = "f64." + ("grad_through_T." if through else "")
        lf = R2REILoss(transform=T, sigma=sigma, no_grad=not through, metric=None)
        l_r2r = lf.r2r_loss(y=y, physics=op, model=net, _unit_noise=n0)
        l_ei = lf.ei_loss(y=y, physics=op, model=net, _n1=n1, _n2=n2)
        assert rel(l_r2r, g[p + "loss_r2r"]) < 1e-4 and rel(l_ei, g[p + "loss_ei"]) < 1e-4
        m.zero_grad_flat()
        total = lf(y=y, physics=op, model=net, _noise=(n0, n1, n2))
        assert rel(total, g[p + "loss"]) < 1e-4
        total.backward()
        for k, q in m.named_parameters():
            if p + f"grad.{k}" in g.files:
                assert relerr(q.grad, g[p + f"grad.{k}"]) < 5e-4, (k, relerr(q.grad, g[p + f"grad.{k}"]))
            else:
                assert rel(q.grad.norm(), g[p + f"gradnorm.{k}"]) < 5e-4, (p, k)
    # (ii) the hot path's EI glue
    with torch.no_grad():
        x1 = m((y + 0.5 * sigma * n1).contiguous())
    assert relerr(x1, g["f64.x1"]) < 2e-5
    op.noise_model = physics.GaussianNoise(sigma=1.5 * sigma)
    ei = EILoss(transform=scaling, metric=mse(), weight=1.0, no_grad=True)
    val = ei(x_net=x1, physics=op, model=net, transform_params=params, noise=n2)
    assert rel(val, g["f64.loss_ei"]) < 1e-4


# ------------------------------------------------------------------ the reference's own loss layer end to end, golden G15
@pytest.mark.parametrize("tag", ["deblur", "sr2"])
def test_loss_layer_vs_reference_classes_golden(golden, tag):
    """G15 = the reference's get_physics / get_loss / Loss.forward / ProposedLoss run by their own code on a seeded CPU
    generator (tools/gen_golden.py gen_loss_glue). The product's get_loss built from the same flags picks the same margin
    and crop ratio, and its loss layer on the GPU -- crop at the offsets that seed drew, the draws that seed handed out
    injected -- gives the reference's float32 loss value and gradient norms."""
    import bench
    import physics as physics_pkg
    from crop import CropPair
    from losses import get_loss
    from losses.sure import embed_probe
    from models.convolutional import ConvolutionalModel
    g = golden(f"g15_loss_glue_{tag}")
    up = 1 if tag == "deblur" else 2
    args = bench.reference_args("cuda", hidden=8, scales=3, task="deblurring" if tag == "deblur" else "sr", sr_factor=None if tag == "deblur" else 2)
    p = physics_pkg.get_physics(args, "cuda")
    lf = get_loss(args, p)
    margin = int(g["sure_margin"])
    assert lf.loss.sure.margin == margin and lf.loss.sure.div_margin == margin and lf.xy_size_ratio == int(g["xy_size_ratio"])
    m = ConvolutionalModel(in_channels=3, upsampling_rate=up, residual=True, inner_residual=True, num_conv_blocks=1,
                           hidden_channels=8, inout_convs=True, scales=3)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    y = dev(g["y"])
    for seed in (0, 1):
        pre = f"f32.seed{seed}."
        i, j = (int(v) for v in g[pre + "ij"])
        yc = CropPair("random", 48).write_y(y, i, j, torch.empty(y.shape[0], 3, 48, 48, device="cuda"))
        draws = {"b": embed_probe(yc, dev(g[pre + "b"]), margin), "rate": dev(g[pre + "rate"]),
                 "center": dev(g[pre + "center"]).view(-1, 1, 1, 2), "noise": dev(g[pre + "n"])}
        m.zero_grad_flat()
        net = lambda v, *ignored: m(v)                       # (src/models/__init__.py:148-149: Model.forward drops the extras)
        val = lf.loss(x=None, y=yc, model=net, draws=draws)
        assert rel(val, g[pre + "loss"]) < 1e-4, (seed, float(val), float(g[pre + "loss"]))
        val.backward()
        for k, q in m.named_parameters():
            assert rel(q.grad.norm(), g[pre + f"gradnorm.{k}"]) < 1e-3, (seed, k)
        # the other methods through the reference's own classes: supervised needs no device draw at all -- the product's
        # Loss.forward from the same CPU seed (its CropPair draws the reference's two offsets) must give the reference's value
        import argparse
        a2 = argparse.Namespace(**vars(args))
        a2.method = "supervised"
        sup = get_loss(a2, p)
        m.zero_grad_flat()
        torch.manual_seed(seed)
        v3 = sup(x=dev(g["x"]), y=y, model=net)
        assert rel(v3, g[pre + "supervised.loss"]) < 1e-4, (seed, "supervised")
        v3.backward()
        got = torch.stack([q.grad.norm() for q in m.parameters()])
        assert relerr(got, g[pre + "supervised.gradnorm"]) < 1e-3
        a2.method = "sure"
        sure = get_loss(a2, p)
        m.zero_grad_flat()
        v2 = sure.loss(x=None, y=yc, model=net, draws={"b": draws["b"]})
        assert rel(v2, g[pre + "sure.loss"]) < 1e-4, (seed, "sure")
        v2.backward()
        got = torch.stack([q.grad.norm() for q in m.parameters()])
        assert relerr(got, g[pre + "sure.gradnorm"]) < 1e-3


def test_fused_and_literal_pass_orders_agree():
    import physics
    import models
    from losses import get_loss
    args = ref_args()
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda").to("cuda")
    x = torch.rand(4, 3, 256, 256, device="cuda")
    y = p(x)
    out = {}
    for fuse in (True, False):
        lf = get_loss(args, p)
        lf.loss.fuse_passes = fuse
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        model.get_backbone().zero_grad_flat()
        val = lf(x=x, y=y, model=model)
        val.backward()
        out[fuse] = (float(val), model.get_backbone().flat_grads.clone())
    assert rel(out[True][0], out[False][0]) < 1e-6
    assert relerr(out[True][1], out[False][1]) < 1e-5


def test_loss_surface_and_methods():
    import physics
    import models
    from losses import get_loss
    args = ref_args()
    p = physics.get_physics(args, "cuda")
    model = models.get_model(args, p, "cuda").to("cuda")
    x = torch.rand(2, 3, 256, 256, device="cuda")
    y = p(x)
    for method in ["proposed", "supervised", "sure", "css", "noise2inverse"]:
        args.method = method
        val = get_loss(args, p)(x=x, y=y, model=model)
        assert val.dim() == 0 and torch.isfinite(val)
    args.method = "nope"
    with pytest.raises(ValueError):
        get_loss(args, p)
    args.method = "proposed"
    args.ProposedLoss__transforms = "Rotations"
    val = get_loss(args, p)(x=x, y=y, model=model)
    assert val.dim() == 0 and torch.isfinite(val)
    args.ProposedLoss__transforms = "Reflections"
    with pytest.raises(ValueError):
        get_loss(args, p)
    # SR: margin 0, x/y size ratio from physics.rate, 48 -> 96 crops
    args = ref_args(task="sr", sr_factor=2, kernel=None)
    p = physics.get_physics(args, "cuda")
    model = models.get_model(args, p, "cuda").to("cuda")
    lf = get_loss(args, p)
    assert lf.loss.sure.margin == 0 and lf.xy_size_ratio == 2
    xs = torch.rand(2, 3, 96, 96, device="cuda")
    val = lf(x=xs, y=p(xs), model=model)
    assert torch.isfinite(val)
    val.backward()


# ------------------------------------------------------------------ full-size network: config[1] parity
@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_default_unet_step_vs_oracle(mode):
    """The benchmarked network (hidden 32, 5 scales, 645 M parameters) at the training crop size:
    restored images within 1e-4 relative and 0.01 dB PSNR of the float32 CPU path; loss and gradient
    norms within 1e-4 (SURVEY 8d 'parity check in the same run'). mode: the exact-f32 MFMA GEMMs, and the split-bf16
    mode (three bf16 MFMA products per float32 product, models/_ops.py gemm_x3) held to the SAME bars."""
    from models import _ops
    prev = _ops.set_compute_dtype(mode)
    try:
        _default_unet_step_vs_oracle(B=2)
    finally:
        _ops.set_compute_dtype(prev)


def test_split_bf16_step_vs_oracle_at_a_batch_whose_weight_gradients_split_too():
    """bf16x3 at batch 8 (72 bottleneck rows: every GEMM of the step, weight gradients included, runs as three bf16
    products -- at batch 2 the 18-row reductions of the deepest weight gradients stay on the float32 GEMM), same bars."""
    from models import _ops
    prev = _ops.set_compute_dtype("bf16x3")
    try:
        _default_unet_step_vs_oracle(B=8)
    finally:
        _ops.set_compute_dtype(prev)


def _default_unet_step_vs_oracle(B, modes=None):
    """modes: run the HIP path once per arithmetic mode against ONE evaluation of the CPU oracle (None: the mode in effect)."""
    import bench
    import metrics
    import models
    import physics
    from losses.sure import embed_probe
    from models import _ops
    args = bench.reference_args("cuda")
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.get_weights().items()}   # CPU, f32
    model.to("cuda")
    gen = torch.Generator().manual_seed(11)
    xgt = torch.rand((B, 3, 48, 48), generator=gen)
    b_int = torch.randn((B, 3, 36, 36), generator=gen)
    noise = torch.randn((B, 3, 48, 48), generator=gen)
    rate = torch.tensor([0.75, 0.5]).repeat(B // 2)
    center = torch.tensor([[0.3, -0.2], [-0.5, 0.6]]).repeat(B // 2, 1)
    k = tp.blur_kernel("Gaussian_R2")
    A = lambda v: tp.blur_fft(v, k)
    y = A(xgt) + 5 / 255 * torch.randn((B, 3, 48, 48), generator=gen)
    ref_model = lambda v: tp.unet_forward(sd, v, scales=5)
    ref, aux = tp.proposed_loss(y, A, ref_model, 5 / 255, margin=6, rate=rate, center=center.view(-1, 1, 1, 2),
                                b=b_int, n=noise)
    ref.backward()
    yd = y.cuda()
    for mode in (modes or [None]):
        prev = _ops.set_compute_dtype(mode) if mode is not None else None
        try:
            loss, got = _proposed_manual(model, p, yd, embed_probe(yd, b_int.cuda(), 6), noise.cuda(), rate.cuda(),
                                         center.cuda(), 6)
            assert relerr(got["x_net"], aux["x_net"]) < 1e-4, mode
            for i in range(B):
                d = abs(float(metrics.psnr_fn(got["x_net"][i].cpu(), xgt[i])) - float(tp.psnr_y(aux["x_net"][i].detach(), xgt[i])))
                assert d < 0.01, mode
            assert rel(loss, ref) < 1e-4, mode
            model.get_backbone().zero_grad_flat()
            loss.backward()
            params = dict(model.get_backbone().named_parameters())
            worst = 0.0
            for kname, v in sd.items():
                gn, rn = float(params[kname].grad.double().norm()), float(v.grad.double().norm())
                worst = max(worst, abs(gn - rn) / rn)
            assert worst < 1e-4, (mode, worst)
        finally:
            if mode is not None:
                _ops.set_compute_dtype(prev)


def test_parity_modes_at_the_timed_batch_vs_oracle():
    """The two parity modes END TO END at the batch bench.py times (B = 32: 576 + 288 bottleneck rows, where the f32 GEMM's
    96 x 128 tiles and the quadrant tiles of the bf16x3 launches exist), against one evaluation of the
    float32 CPU oracle: restored images 1e-4 / 0.01 dB, loss 1e-4, every parameter-gradient norm 1e-4."""
    _default_unet_step_vs_oracle(B=32, modes=["f32", "bf16x3"])


def _launch_plans(records):
    """{(family, tile rows, tile columns)} of the sei_gemm_bf16nt launches in a recorded step (sei_gemm_bf16nt_plan)."""
    import _native
    plans = {}
    for _, entry, a in records:
        if entry == "sei_gemm_bf16nt_colsum":               # (bf16 result + its column sums: the same dispatch)
            M, Nn, K, epi = a[7:11]
            fam, bm, bn, sk = _native.gemm_plan(a[2], a[5], False, True, M, Nn, K, epi)
            plans.setdefault((fam, bm, bn), set()).add((M, Nn, K, sk))
            continue
        if entry == "sei_gemm_bf16nt_ws":                   # (split-K workspace: slices meet in slabs where they split)
            M, Nn, K, epi = a[8:12]
            fam, bm, bn, sk, _ = _native.gemm_plan(a[2], a[5], bool(a[6]), bool(a[7]), M, Nn, K, epi, ws_bytes=a[18])
            plans.setdefault((fam, bm, bn), set()).add((M, Nn, K, sk))
            continue
        if entry != "sei_gemm_bf16nt":
            continue
        M, Nn, K, epi = a[8:12]
        fam, bm, bn, sk = _native.gemm_plan(a[2], a[5], bool(a[6]), bool(a[7]), M, Nn, K, epi)
        plans.setdefault((fam, bm, bn), set()).add((M, Nn, K, sk))
    return plans


@pytest.mark.parametrize("B,oracle_dtype", [(8, torch.float64), (32, torch.float32)])
def test_timed_configuration_vs_oracle(B, oracle_dtype):
    """The configuration bench.py times, INCLUDING its optimizer step -- bf16 GEMMs, hipGraph replay, the fused 2B pass,
    merged and STORED weight gradients, the 1x1 convolution behind the ideal downsampler, the default 645 M-parameter
    network (hidden 32, 5 scales) and, at B = 8 (72 + 144 bottleneck rows: the merged launch the bench times exists),
    torch.optim.Adam applied to the two deepest levels' weights in the epilogue of their weight-gradient GEMMs
    (`fuse_optimizer=True`, demo/train.py:262-268) -- value-pinned against the float64 oracle on the same weights, crop
    and injected randomness: restored images within 0.01 dB PSNR-Y, loss within bf16 rounding, every parameter gradient
    aligned with the oracle's (cosine; for the fused weights the gradient is read back from exp_avg = 0.1 g after the
    first step), and the POST-STEP parameters / moments against the oracle's torch.optim.Adam step in float64.

    B = 32 is the batch bench.py TIMES (576 + 288 bottleneck rows): the 288-row quadrant tiles and the split-K quadrant
    launches only exist there (at B = 8 the bottleneck GEMMs have 144 / 72 rows and take 128 x 128 tiles), so the same
    bars are held at 32 against the float32 oracle (the float64 one would take ~4 minutes) -- and the test asserts,
    through sei_gemm_bf16nt_plan, that the recorded launch set of the step under test really contains those schedules:
    a dispatch change cannot silently move the benchmarked kernels away from what is pinned here."""
    import bench
    import metrics
    import models
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from losses.sure import embed_probe
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda")
        p = physics.get_physics(args, "cuda")
        torch.manual_seed(0)
        model = models.get_model(args, p, "cuda")
        sd = {k: v.detach().to(oracle_dtype).requires_grad_(True) for k, v in model.get_weights().items()}   # CPU
        model.to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        lf.loss.keep_outputs = True
        lr = 1e-4
        opt = FlatAdam(model, lr=lr)
        gen = torch.Generator().manual_seed(11)
        x = torch.rand((B, 3, 256, 256), generator=gen)
        k = tp.blur_kernel("Gaussian_R2")
        A = lambda v: tp.blur_fft(v, k)
        y = A(x) + 5 / 255 * torch.randn((B, 3, 256, 256), generator=gen)
        b_int = torch.randn((B, 3, 36, 36), generator=gen)
        noise = torch.randn((B, 3, 48, 48), generator=gen)
        rate = torch.tensor([0.75, 0.5, 0.5, 0.75, 0.75, 0.5, 0.75, 0.5] * (B // 8))
        center = 2 * torch.rand((B, 2), generator=gen) - 1
        graphed = GraphedLossStep(lf, model, opt, (B, 3, 48, 48), fuse_optimizer=True)
        assert graphed.store_weight_grads and bb._sei_zero_ranges is not None
        assert len(graphed.fused_views) == 8                     # levels 3 and 4: 98.8 % of the parameters
        fused_ranges = list(opt._fused_ranges)
        assert sum(hi - lo for lo, hi in fused_ranges) == 2 * 8192 * 32768 + 6 * 2048 * 8192
        xd, yd = x.cuda(), y.cuda()
        draws = {"b": embed_probe(graphed.static_y, b_int.cuda(), 6), "rate": rate.cuda(), "center": center.cuda(),
                 "noise": noise.cuda()}
        st = opt.state[bb.flat_params]
        start = bb.flat_params.clone()
        # the launch set of this step (one eager twin step, recorded; its weights are rewound below)
        _ops.profile_gemms(True)
        try:
            _ops.set_fused_adam(*graphed.fused_table, owner=bb)
            opt.prepare_step()
            bb.zero_grad_flat(store_weight_grads=True)
            torch.manual_seed(5)
            lf(x=xd, y=yd, model=model, draws=draws).backward()
        finally:
            _ops.set_fused_adam(None, None, owner=bb)
            plans = _launch_plans(_ops.profile_gemms(False))
        if B == 32:
            assert ("pq", 288, 256) in plans and ("pq", 288, 128) in plans, sorted(plans)
            assert any(sk > 1 for key in plans if key[0] == "pq" for *_, sk in plans[key]), "split-K quadrant launches"
            assert {(576, 32768, 8192, 1), (576, 8192, 32768, 4), (288, 8192, 32768, 7)} <= plans[("pq", 288, 256)]
            assert {(288, 32768, 8192, 1), (2304, 2048, 8192, 2), (1152, 2048, 8192, 4)} <= plans[("pq", 288, 128)]
        else:
            assert any(M in (72, 144) and Nn == 32768 for M, Nn, _, _ in plans[("nt", 128, 128)]), sorted(plans)
        print("launch plans:", {k: len(v) for k, v in sorted(plans.items())})
        for _ in range(2):                                  # the second pass is the one checked (stale state shows):
            bb.flat_params.copy_(start)                     # every replay steps the deep weights, so rewind first
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
            st["step"] = 0
            _ops.refresh_plain_shadow(bb)
            bb.flat_grads.fill_(float("nan"))
            torch.manual_seed(5)                            # the crop offsets
            loss = float(graphed(xd, yd, draws=draws))
            opt.step()
        torch.cuda.synchronize()
        assert st["step"] == 1
        x_net = lf.loss.kept["x_net"].float().cpu()
        torch.manual_seed(5)
        xc, yc = tp.crop_pair(x, y, 48, 1)
        ref_model = lambda v: tp.unet_forward(sd, v, scales=5)
        od = oracle_dtype
        ref, aux = tp.proposed_loss(yc.to(od), lambda v: tp.blur_fft(v, k), ref_model, 5 / 255, margin=6,
                                    rate=rate.to(od), center=center.to(od).view(-1, 1, 1, 2), b=b_int.to(od),
                                    n=noise.to(od))
        ref.backward()
        assert torch.equal(graphed.static_y.cpu(), yc.contiguous())
        for i in range(B):
            d = abs(float(metrics.psnr_fn(x_net[i], xc[i])) - float(tp.psnr_y(aux["x_net"][i].detach(), xc[i].to(od))))
            assert d < 0.01, d
        assert relerr(x_net, aux["x_net"]) < 2e-2
        assert abs(loss - float(ref)) < 2e-2 * abs(float(ref)), (loss, float(ref))
        # gradients: the stored ones from the bucket, the fused ones from the first moment (exp_avg = (1 - beta1) g).
        # (The per-parameter comparisons run on the GPU in float64: the oracle's values are only COPIED there.)
        base, esz = bb.flat_grads.data_ptr(), 4
        in_fused = lambda off: any(lo <= off < hi for lo, hi in fused_ranges)
        cosine = lambda u, v: float(u @ v / (u.norm() * v.norm()))
        worst_big, worst_small, n_fused = 1.0, 1.0, 0
        for name, prm in bb.named_parameters():
            off = (prm._sei_grad_view.data_ptr() - base) // esz
            if in_fused(off):
                g = st["exp_avg"][off:off + prm.numel()].double() / 0.1
                n_fused += 1
            else:
                g = prm.grad.double().flatten()
                assert torch.isfinite(g).all(), name
            r = sd[name].grad.flatten().cuda().double()
            cos = cosine(g, r)
            if prm.dim() == 4 and prm.shape[-1] == 1 and prm.numel() >= 4096:
                worst_big = min(worst_big, cos)
            else:
                worst_small = min(worst_small, cos)
            assert cos > 0.99, (name, cos)
            assert abs(float(g.norm() / r.norm()) - 1) < 2e-2, (name, float(g.norm()), float(r.norm()))
        assert n_fused == 8 and worst_big > 0.999, (n_fused, worst_big)
        # the optimizer step itself: torch.optim.Adam in float64 on the oracle's gradients (demo/train.py:157-186,266-268)
        ref_opt = torch.optim.Adam(list(sd.values()), lr=lr, betas=(0.9, 0.999))
        before = {n: v.detach().clone() for n, v in sd.items()}
        ref_opt.step()
        worst_m, worst_v, worst_dp, flips = 1.0, 1.0, 1.0, 0.0
        for name, prm in bb.named_parameters():
            off = (prm.data_ptr() - bb.flat_params.data_ptr()) // 4
            n = prm.numel()
            rs = ref_opt.state[sd[name]]
            m = st["exp_avg"][off:off + n].double()
            v = st["exp_avg_sq"][off:off + n].double()
            rm, rv = rs["exp_avg"].flatten().cuda().double(), rs["exp_avg_sq"].flatten().cuda().double()
            b4 = before[name].flatten().cuda().double()
            dp = prm.detach().double().flatten() - b4
            rdp = sd[name].detach().flatten().cuda().double() - b4
            # the first Adam step moves every weight by lr * g / (|g| + eps): at most lr, whatever the gradient
            assert float(dp.abs().max()) <= lr * (1 + 1e-3) and float((dp - rdp).abs().max()) <= 2 * lr * (1 + 1e-3), name
            big = prm.dim() == 4 and prm.shape[-1] == 1 and n >= 4096
            cm, cv, cdp = cosine(m, rm), cosine(v, rv), cosine(dp, rdp)
            assert cm > 0.99 and cv > 0.98 and cdp > 0.9, (name, cm, cv, cdp)
            if big:
                worst_m, worst_v, worst_dp = min(worst_m, cm), min(worst_v, cv), min(worst_dp, cdp)
                # where the oracle's gradient is not within bf16 noise of zero, the step has the oracle's sign
                sure_sign = rm.abs() > 0.1 * rm.abs().mean()
                flips = max(flips, float((torch.sign(dp[sure_sign]) != torch.sign(rdp[sure_sign])).double().mean()))
            del m, v, rm, rv, b4, dp, rdp
        assert worst_m > 0.999 and worst_v > 0.998, (worst_m, worst_v)
        assert worst_dp > 0.97 and flips < 0.02, (worst_dp, flips)
        # the bf16 shadow the next forward reads is the rounded new parameter
        assert torch.equal(bb.flat_shadow, bb.flat_params.bfloat16())
        print(f"timed configuration (B = {B}, Adam inside the deep weight-gradient GEMMs) vs {od} oracle: loss {loss:.6f} vs "
              f"{float(ref):.6f}; gradient cosine >= {worst_big:.5f} (1x1 weights), >= {worst_small:.5f} (others); "
              f"post-step exp_avg cosine >= {worst_m:.5f}, exp_avg_sq >= {worst_v:.5f}, parameter-step cosine >= "
              f"{worst_dp:.4f}, sign flips on non-negligible gradients <= {flips:.4f}")
    finally:
        _ops.set_compute_dtype(prev)


@pytest.mark.parametrize("net", ["h8s3_crop48", "default_crop16"])
def test_sr4_composite_step_vs_oracle(net):
    """BASELINE configs[2]: one proposed-loss step for super-resolution x4 in float32 -- Loss.forward's paired crop
    (x at 4x the offsets, the batched-padding quirk), SURE with margin 0 through the antialiased x4 downsampling
    (src/physics/downsampling/__init__.py:16-19), the x4 pre-upsampler of the U-Net, the scale transform on the
    192x192 estimate -- against the oracle's restatement of src/losses/__init__.py:133-142,222-226 with the same
    weights, crop and injected randomness."""
    import models
    import physics
    from losses import get_loss
    hidden, scales, crop = (8, 3, 48) if net == "h8s3_crop48" else (32, 5, 16)
    args = ref_args(task="sr", sr_factor=4, kernel=None, ConvolutionalModel__hidden_channels=hidden,
                    ConvolutionalModel__scales=scales, Loss__crop_size=crop)
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.get_weights().items()}
    model.to("cuda")
    lf = get_loss(args, p)
    assert lf.loss.sure.margin == 0 and lf.xy_size_ratio == 4
    gen = torch.Generator().manual_seed(3)
    B = 2
    x = torch.rand((B, 3, 4 * crop, 4 * crop), generator=gen)
    y = tp.downsample_aa(x, 4) + 5 / 255 * torch.randn((B, 3, crop, crop), generator=gen)
    b = torch.randn((B, 3, crop, crop), generator=gen)
    noise = torch.randn((B, 3, crop, crop), generator=gen)
    rate, center = torch.tensor([0.5, 0.75]), torch.tensor([[-0.4, 0.1], [0.7, -0.6]])
    torch.manual_seed(9)
    model.get_backbone().zero_grad_flat()
    val = lf(x=x.cuda(), y=y.cuda(), model=model,
             draws={"b": b.cuda(), "rate": rate.cuda(), "center": center.cuda().view(B, 1, 1, 2), "noise": noise.cuda()})
    val.backward()
    torch.manual_seed(9)
    _, yc = tp.crop_pair(x, y, crop, 4)
    ref, aux = tp.proposed_loss(yc.contiguous(), lambda v: tp.downsample_aa(v, 4),
                                lambda v: tp.unet_forward(sd, v, scales=scales, upsampling_rate=4), 5 / 255, margin=0,
                                rate=rate, center=center.view(B, 1, 1, 2), b=b, n=noise)
    ref.backward()
    assert aux["x_net"].shape == (B, 3, 4 * crop, 4 * crop)
    assert rel(val, ref) < 1e-4, (float(val), float(ref))
    params = dict(model.get_backbone().named_parameters())
    worst = 0.0
    for name, v in sd.items():
        gn, rn = float(params[name].grad.double().norm()), float(v.grad.double().norm())
        worst = max(worst, abs(gn - rn) / rn)
    assert worst < 1e-4, worst


@pytest.mark.parametrize("crop,B,oracle_dtype", [(16, 2, torch.float64), (48, 1, torch.float32)])
def test_sr4_bf16_graphed_step_vs_oracle(crop, B, oracle_dtype):
    """BASELINE configs[2] as it is benchmarked (`bench.py --task sr`, bf16): the default 645 M-parameter network with its
    x4 pre-upsampler, x4 antialiased downsampling physics, SURE margin 0, paired crop, bf16 GEMMs, hipGraph replay with
    merged + stored weight gradients -- against the FLOAT64 oracle on the same weights, crop (16: the network runs at
    64x64) and injected draws: restored images within 0.01 dB PSNR-Y, loss within bf16 rounding, gradient cosines.
    (The float64 oracle of this 645 M-parameter network is ~1 minute of host time on the box's 16 CPUs; the float32
    twin of this step is test_sr4_composite_step_vs_oracle.)
    Second case: THE SIZE THAT IS TIMED -- crop 48, the network on a 192 x 192 grid (the dispatch branches of that size:
    lane-per-channel 3 -> 32 data gradient, the 24-output resampler maps, the large-grid depthwise kernels, weight
    gradients over 36,864-pixel reductions) -- one pair, against the float32 oracle (~3 TFLOP of host work)."""
    import bench
    import metrics
    import models
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        od = oracle_dtype
        args = ref_args(task="sr", sr_factor=4, kernel=None, Loss__crop_size=crop, ConvolutionalModel__hidden_channels=32,
                        ConvolutionalModel__scales=5)
        p = physics.get_physics(args, "cuda")
        torch.manual_seed(0)
        model = models.get_model(args, p, "cuda")
        sd = {k: v.detach().to(od).clone().requires_grad_(True) for k, v in model.get_weights().items()}
        model.to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        lf.loss.keep_outputs = True
        assert lf.loss.sure.margin == 0 and lf.xy_size_ratio == 4
        opt = FlatAdam(model, lr=1e-4)
        gen = torch.Generator().manual_seed(3)
        x = torch.rand((B, 3, 4 * crop, 4 * crop), generator=gen)
        y = tp.downsample_aa(x, 4) + 5 / 255 * torch.randn((B, 3, crop, crop), generator=gen)
        b = torch.randn((B, 3, crop, crop), generator=gen)
        noise = torch.randn((B, 3, crop, crop), generator=gen)
        rate, center = torch.tensor([0.5, 0.75])[:B], torch.tensor([[-0.4, 0.1], [0.7, -0.6]])[:B]
        graphed = GraphedLossStep(lf, model, opt, (B, 3, crop, crop))
        assert graphed.store_weight_grads
        draws = {"b": b.cuda(), "rate": rate.cuda(), "center": center.cuda(), "noise": noise.cuda()}
        xd, yd = x.cuda(), y.cuda()
        for _ in range(2):
            bb.flat_grads.fill_(float("nan"))
            torch.manual_seed(9)
            loss = float(graphed(xd, yd, draws=draws))
        x_net = lf.loss.kept["x_net"].float().cpu()
        torch.manual_seed(9)
        xc, yc = tp.crop_pair(x, y, crop, 4)
        assert torch.equal(graphed.static_y.cpu(), yc.contiguous())
        ref, aux = tp.proposed_loss(yc.contiguous().to(od), lambda v: tp.downsample_aa(v, 4),
                                    lambda v: tp.unet_forward(sd, v, scales=5, upsampling_rate=4), 5 / 255, margin=0,
                                    rate=rate.to(od), center=center.to(od).view(B, 1, 1, 2), b=b.to(od),
                                    n=noise.to(od))
        ref.backward()
        assert aux["x_net"].shape == (B, 3, 4 * crop, 4 * crop) == x_net.shape
        for i in range(B):
            d = abs(float(metrics.psnr_fn(x_net[i], xc[i])) - float(tp.psnr_y(aux["x_net"][i].detach().double(), xc[i].double())))
            assert d < 0.01, d
        assert relerr(x_net, aux["x_net"]) < 2e-2
        assert abs(loss - float(ref)) < 2e-2 * abs(float(ref)), (loss, float(ref))
        assert torch.isfinite(bb.flat_grads).all()
        worst_big, worst_small = 1.0, 1.0
        for name, prm in bb.named_parameters():
            g, r = prm.grad.double().flatten().cpu(), sd[name].grad.double().flatten()
            cos = float(g @ r / (g.norm() * r.norm()))
            if prm.dim() == 4 and prm.shape[-1] == 1 and prm.numel() >= 4096:
                worst_big = min(worst_big, cos)
            else:
                worst_small = min(worst_small, cos)
            assert cos > 0.99, (name, cos)
        assert worst_big > 0.999, worst_big
        print(f"SR x4 bf16 graphed step (crop {crop}, B {B}) vs {str(od)[6:]} oracle: loss {loss:.6f} vs {float(ref):.6f}; gradient cosine >= "
              f"{worst_big:.5f} (1x1 weights), >= {worst_small:.5f} (others)")
    finally:
        _ops.set_compute_dtype(prev)


@pytest.mark.parametrize("method", ["supervised", "css", "sure"])
def test_other_methods_vs_oracle(method):
    """supervised / css: mean((model(y) - x)^2) on the cropped pair (src/losses/__init__.py:13-46; css differs in
    the dataset, not in the loss); sure: src/losses/__init__.py:49-64 with the probe injected. Loss value and
    every parameter gradient against the oracle in float32."""
    import models
    import physics
    import torch.nn.functional as F
    from losses import get_loss
    from losses.sure import embed_probe
    args = ref_args(method=method)
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda")
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.get_weights().items()}
    model.to("cuda")
    lf = get_loss(args, p)
    gen = torch.Generator().manual_seed(4)
    x = torch.rand((3, 3, 256, 256), generator=gen)
    k = tp.blur_kernel("Gaussian_R2")
    A = lambda v: tp.blur_fft(v, k)
    y = A(x) + 5 / 255 * torch.randn((3, 3, 256, 256), generator=gen)
    b_int = torch.randn((3, 3, 36, 36), generator=gen)
    torch.manual_seed(13)
    xc, yc = tp.crop_pair(x, y, 48, 1)
    net = lambda v: tp.unet_forward(sd, v, scales=3)
    if method == "sure":
        ref = tp.sure_loss(yc, net(yc), A, net, 5 / 255, margin=6, b=b_int)
        draws = {"b": embed_probe(yc.cuda().contiguous(), b_int.cuda(), 6)}
    else:
        ref = F.mse_loss(net(yc), xc)
        draws = None
    ref.backward()
    torch.manual_seed(13)
    model.get_backbone().zero_grad_flat()
    val = lf(x=x.cuda(), y=y.cuda(), model=model, draws=draws)
    val.backward()
    assert rel(val, ref) < 1e-4, (float(val), float(ref))
    for name, prm in model.get_backbone().named_parameters():
        assert relerr(prm.grad, sd[name].grad) < 1e-3, (name, relerr(prm.grad, sd[name].grad))


@pytest.mark.parametrize("flags,graphed", [
    (["--method", "supervised"], True), (["--method", "css"], True), (["--method", "sure"], True),
    (["--method", "noise2inverse"], True),
    (["--method", "proposed", "--ProposedLoss__transforms", "Shifts"], False),
    (["--method", "proposed", "--ProposedLoss__transforms", "Rotations+Shifts"], False),
    (["--method", "proposed", "--ScalingTransform__kind", "normal"], False),
    (["--method", "proposed", "--ScalingTransform__antialias", "--batch_size", "1"], False),
    # the split-bf16 parity mode through the driver (hidden 32: its GEMM shapes take gemm_x3; captured like the f32 mode)
    (["--method", "proposed", "--compute_dtype", "bf16x3", "--ConvolutionalModel__hidden_channels", "32"], True)])
def test_train_script_methods_and_transforms(tmp_path, flags, graphed):
    """train.py with default --hip_graph for every supported method / transform: steps that draw on the host or
    sync with it (Shifts, the normal kind, the antialiased variant) must fall back to the eager step instead of
    being captured (graphs.can_capture); the others replay a graph with a static x where the loss reads it."""
    out = tmp_path / "run"
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--device", "cuda", "--task", "deblurring", "--kernel",
           "Gaussian_R2", "--ProposedModel__architecture", "Convolutional", "--ConvolutionalModel__hidden_channels",
           "8", "--ConvolutionalModel__scales", "3", "--dataset", "synthetic", "--batch_size", "2", "--epochs", "4",
           "--max_steps", "2", "--out_dir", str(out)] + flags
    env = dict(os.environ, SEI_TRACE_STEP_KIND="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = open(out / "training.csv").read().strip().splitlines()
    assert len(rows) == 5 and all(np.isfinite(float(r_.split(",")[1])) for r_ in rows[1:])
    assert ("step kind: hipGraph replay" in r.stdout) == graphed, r.stdout
    assert ("step kind: eager" in r.stdout) == (not graphed), r.stdout


# ------------------------------------------------------------------ drivers
def test_train_script_end_to_end(tmp_path):
    out = tmp_path / "run"
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--device", "cuda", "--method", "proposed", "--task",
           "deblurring", "--kernel", "Gaussian_R2", "--ProposedModel__architecture", "Convolutional",
           "--ConvolutionalModel__hidden_channels", "8", "--ConvolutionalModel__scales", "3", "--dataset",
           "synthetic", "--batch_size", "4", "--epochs", "4", "--max_steps", "2", "--checkpoint_interval", "2",
           "--out_dir", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = open(out / "training.csv").read().strip().splitlines()
    assert rows[0] == "Epoch,Training Loss" and len(rows) == 5
    assert all(np.isfinite(float(r_.split(",")[1])) for r_ in rows[1:])
    names = sorted(os.listdir(out / "checkpoints"))
    assert names == ["ckp_0.pt", "ckp_1.pt", "ckp_3.pt", "ckp_4.pt"]
    ckp = torch.load(out / "checkpoints" / "ckp_4.pt", map_location="cpu")
    assert set(ckp) == {"epoch", "params", "optimizer", "scheduler"} and ckp["epoch"] == 3
    w = torch.load(out / "weights.pt", map_location="cpu")
    assert "seq.0.in_conv.weight" in w and set(w) == set(ckp["params"])
    # resume: loads model/optimizer/scheduler, requires --lr (as upstream)
    r = subprocess.run(cmd[:-1] + [str(tmp_path / "run2"), "--RESUME", str(out / "checkpoints" / "ckp_4.pt"),
                                   "--lr", "1e-5", "--epochs", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_r2r_alternative_and_shift_transform_match_oracle():
    """--ProposedLoss__sure_alternative r2r with --ProposedLoss__transforms Shifts: loss and weight gradients
    against the oracle's restatement of src/losses/r2r.py on the same weights, shifts and injected noise."""
    import physics
    import models
    import transforms
    from losses import get_loss
    args = ref_args(ProposedLoss__sure_alternative="r2r", ProposedLoss__transforms="Shifts",
                    ConvolutionalModel__hidden_channels=8, ConvolutionalModel__scales=3)
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda").to("cuda")
    lf = get_loss(args, p)
    inner = lf.loss.loss_fns[0]
    assert isinstance(inner.T, transforms.Shift) and lf.loss.compute_x_net is False
    gen = torch.Generator().manual_seed(5)
    y = torch.rand((2, 3, 48, 48), generator=gen)
    n0, n1, n2 = (torch.randn((2, 3, 48, 48), generator=gen) for _ in range(3))
    torch.manual_seed(77)                                   # the Shift transform's two randperm draws
    val = inner(y=y.cuda(), physics=p, model=model, x=None, x_net=None, _noise=(n0.cuda(), n1.cuda(), n2.cuda()))
    val.backward()
    sd = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.get_weights().items()}
    k2 = tp.blur_kernel("Gaussian_R2")
    A = lambda v: tp.blur_fft(v, k2)
    net = lambda v: tp.unet_forward(sd, v, scales=3)
    torch.manual_seed(77)
    ref, _, _ = tp.r2r_ei_loss(y.double(), A, net, transforms.Shift(), 5 / 255, unit_pert=n0.double(),
                               n1=n1.double(), n2=n2.double())
    ref.backward()
    assert abs(float(val) - float(ref)) < 1e-4 * abs(float(ref))
    params = dict(model.get_backbone().named_parameters())
    worst = max(abs(float(params[k].grad.double().norm()) - float(v.grad.norm())) / float(v.grad.norm())
                for k, v in sd.items())
    assert worst < 1e-3, worst
    # the Shift transform alone: a circular roll of the whole batch by one (dx, dy)
    t = transforms.Shift()
    x = torch.arange(2 * 3 * 6 * 5, dtype=torch.float32).view(2, 3, 6, 5)
    torch.manual_seed(1)
    out = t(x)
    torch.manual_seed(1)
    sx = int(torch.arange(-6, 6)[torch.randperm(12)][0]); sy = int(torch.arange(-5, 5)[torch.randperm(10)][0])
    assert torch.equal(out, torch.roll(x, [sx, sy], [-2, -1]))


@pytest.mark.parametrize("shape", [(2, 3, 48, 48), (1, 2, 37, 53)])
def test_rotate_transform_vs_oracle(shape):
    """sei_rotate_nearest_fwd/bwd against the oracle's restatement of torchvision's rotate (nearest, zero fill): equal
    wherever the source coordinate is not within 1e-4 of a rounding tie (there float32 evaluation order decides);
    the backward is the exact transpose of the forward map."""
    import transforms
    gen = torch.Generator().manual_seed(21)
    x = torch.rand(shape, generator=gen)
    g = torch.randn(shape, generator=gen)
    t = transforms.Rotate()
    for angle in (1.0, 37.0, 45.0, 90.0, 123.0, 180.0, 270.0, 359.0):
        xd = x.cuda().requires_grad_(True)
        out = t(xd, params=[angle])
        ref = tp.rotate_nearest(x, angle)
        differs = (out.cpu() != ref).any(dim=0).any(dim=0)
        margin = tp.rotate_source_margin(shape[-2:], angle)
        assert not bool((differs & (margin > 1e-4)).any()), angle
        assert float(differs.float().mean()) < 0.01, angle
        out.backward(g.cuda())
        lhs = float((out.detach().double() * g.cuda().double()).sum())
        rhs = float((xd.grad.double() * x.cuda().double()).sum())
        assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs)), (angle, lhs, rhs)     # f32 sums where outputs share a source
        if not bool(differs.any()):
            xr = x.clone().requires_grad_(True)
            tp.rotate_nearest(xr, angle).backward(g)
            assert torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-6), angle
    assert torch.equal(t(x.cuda(), params=[90.0]).cpu(), torch.rot90(x, 1, (-2, -1))) if shape[-1] == shape[-2] else True
    torch.manual_seed(3)
    drawn = t.sample()
    torch.manual_seed(3)
    assert drawn == [float(torch.arange(0, 360)[1:][torch.randperm(359)][0])] and 1.0 <= drawn[0] <= 359.0


@pytest.mark.parametrize("which", ["Rotations", "Rotations+Shifts"])
def test_equivariant_loss_with_rotations_vs_oracle(which):
    """--ProposedLoss__transforms Rotations / Rotations+Shifts (src/losses/__init__.py:84-91): the EI term on the same
    weights, angle, shifts and injected measurement noise as the oracle; value and every weight-gradient norm."""
    import physics
    import models
    import transforms
    from losses import get_loss
    args = ref_args(ProposedLoss__transforms=which)
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda").to("cuda")
    lf = get_loss(args, p)
    ei = lf.loss.ei
    kinds = [type(v) for v in ei.T.transforms] if which.endswith("Shifts") else [type(ei.T)]
    assert kinds == ([transforms.Rotate, transforms.Shift] if which.endswith("Shifts") else [transforms.Rotate])
    assert not lf.loss.graph_safe
    gen = torch.Generator().manual_seed(6)
    y = torch.rand((2, 3, 48, 48), generator=gen)
    n = torch.randn((2, 3, 48, 48), generator=gen)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.get_weights().items()}
    net = lambda v: tp.unet_forward(sd, v, scales=3)
    k2 = tp.blur_kernel("Gaussian_R2")
    A = lambda v: tp.blur_fft(v, k2)
    for seed in range(77, 177):                       # a draw whose sources all sit clear of a rounding tie
        torch.manual_seed(seed)
        if float(tp.rotate_source_margin((48, 48), transforms.Rotate().sample()[0]).min()) > 2e-4:   # float32 coordinate error is ~1e-5
            break
    else:
        raise AssertionError("no tie-free angle among 100 seeds")
    shift = transforms.Shift()

    def oracle_transform(t):
        out = tp.rotate_nearest(t, transforms.Rotate().sample()[0])
        return shift(out) if which.endswith("Shifts") else out

    torch.manual_seed(seed)
    ref, _, _ = tp.ei_loss(net(y), A, net, oracle_transform, 5 / 255, n=n)
    ref.backward()
    model.get_backbone().zero_grad_flat()
    torch.manual_seed(seed)
    val = ei(x_net=model(y.cuda()), physics=p, model=model, noise=n.cuda())
    val.backward()
    assert rel(val, ref) < 1e-4, (float(val), float(ref))
    for name, prm in model.get_backbone().named_parameters():
        assert relerr(prm.grad, sd[name].grad) < 2e-3, (name, relerr(prm.grad, sd[name].grad))


def test_psnr_y_metric_and_registration():
    """metrics.psnr_fn on the GPU (sei_luma_sqerr) against the oracle's luma PSNR; centre-crop registration."""
    import metrics
    gen = torch.Generator().manual_seed(9)
    x = torch.rand((3, 97, 131), generator=gen)
    x_hat = (x + 0.05 * torch.randn((3, 97, 131), generator=gen)).clamp(0, 1)
    ref = float(tp.psnr_y(x_hat.double(), x.double()))
    assert abs(float(metrics.psnr_fn(x_hat.cuda(), x.cuda())) - ref) < 1e-4
    assert abs(float(metrics.psnr_fn(x_hat, x)) - ref) < 1e-4                  # host path (CPU tensors)
    a, b = metrics.register_fn(torch.zeros(3, 100, 90), torch.zeros(3, 96, 95))
    assert a.shape == b.shape == (3, 96, 90)
    psnr, ssim, lpips = metrics.compute_metrics(x.cuda(), x_hat.cuda())
    assert abs(psnr - ref) < 1e-4 and np.isnan(ssim) and np.isnan(lpips)


def test_train_on_files_then_evaluate(tmp_path):
    """N1 + N3 end to end: train.py on a single-image PNG dataset (decode, antialiased resize, seeded
    measurements, random crops), then test.py on a small DIV2K validation tree with the saved weights."""
    from PIL import Image
    rng = np.random.default_rng(11)
    img = tmp_path / "one.png"
    Image.fromarray(rng.integers(0, 256, size=(300, 340, 3), dtype=np.uint8)).save(img)
    common = ["--device", "cuda", "--task", "deblurring", "--kernel", "Gaussian_R2", "--ProposedModel__architecture",
              "Convolutional", "--ConvolutionalModel__hidden_channels", "8", "--ConvolutionalModel__scales", "3"]
    out = tmp_path / "run"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), *common, "--method", "proposed", "--dataset",
                        "single_image", "--SingleImageDataset__image_path", str(img),
                        "--SingleImageDataset__duplicates_count", "8", "--batch_size", "4", "--epochs", "4",
                        "--out_dir", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = open(out / "training.csv").read().strip().splitlines()
    assert len(rows) == 5 and all(np.isfinite(float(r_.split(",")[1])) for r_ in rows[1:])
    # the same run fed from the GPU-resident pair cache
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), *common, "--method", "proposed", "--dataset",
                        "single_image", "--SingleImageDataset__image_path", str(img),
                        "--SingleImageDataset__duplicates_count", "8", "--batch_size", "4", "--epochs", "4",
                        "--device_cache", "--out_dir", str(tmp_path / "run_cache")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Device cache: 8 pairs" in r.stdout, r.stdout + r.stderr
    rows_c = open(tmp_path / "run_cache" / "training.csv").read().strip().splitlines()
    assert len(rows_c) == 5 and all(np.isfinite(float(r_.split(",")[1])) for r_ in rows_c[1:])
    val = tmp_path / "data" / "DIV2K" / "DIV2K_valid_HR"
    val.mkdir(parents=True)
    for k, (h, w) in enumerate([(288, 300), (270, 256)]):
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(val / f"{801 + k:04d}.png")
    res = tmp_path / "eval"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), *common, "--dataset", "div2k",
                        "--GroundTruthDataset__datasets_dir", str(tmp_path / "data"), "--weights",
                        str(out / "weights.pt"), "--indices", "0,1", "--print_all_metrics", "--save_images",
                        "--save_psf", "--out_dir", str(res)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert "N: 2" in lines
    psnr = float([ln for ln in lines if ln.startswith("PSNR:")][0].split()[1])
    per = [float(ln.split("PSNR:")[1].split(",")[0]) for ln in lines if ln.startswith("METRICS_")]
    assert len(per) == 2 and np.isfinite(psnr) and abs(np.mean(per) - psnr) < 0.011 and 3.0 < psnr < 60.0
    for folder in ("ground_truth", "predictors", "estimates"):
        assert sorted(os.listdir(res / folder)) == ["0.png", "1.png"]
    assert os.path.exists(res / "psf.png")
    # the saved estimate reproduces the printed PSNR (8-bit images, luma PSNR of the oracle)
    def load(p_):
        return torch.from_numpy(np.asarray(Image.open(p_), dtype=np.float64).transpose(2, 0, 1) / 255.0)
    again = float(tp.psnr_y(load(res / "estimates" / "0.png"), load(res / "ground_truth" / "0.png")))
    assert abs(again - per[0]) < 0.011
    # the sliced Noise2Inverse evaluation and the R2R average around the same backbone (demo/test.py:116-134)
    for flags in (["--noise2inverse"], ["--r2r", "--r2r_itercount", "2"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "test.py"), *common, "--dataset", "div2k",
                            "--GroundTruthDataset__datasets_dir", str(tmp_path / "data"), "--weights",
                            str(out / "weights.pt"), "--indices", "1", *flags], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "N: 1" in r.stdout, r.stdout + r.stderr
        assert np.isfinite(float([ln for ln in r.stdout.splitlines() if ln.startswith("PSNR:")][0].split()[1]))


def test_graft_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


def test_graphed_step_matches_eager():
    """hipGraph replay of zero_grad + loss + backward == the eager path: the step's device-side draws are made
    eagerly, by the same calls in the same order, into static buffers the graph reads, so equal seeds give equal
    numbers and the comparison is an equality (up to the arrival order of the float atomics)."""
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from optim import FlatAdam
    args = ref_args()
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda").to("cuda")
    lf = get_loss(args, p)
    opt = FlatAdam(model, lr=1e-4)
    x = torch.rand(4, 3, 256, 256, device="cuda")
    y = p(x)
    graphed = GraphedLossStep(lf, model, opt, (4, 3, 48, 48))
    bb = model.get_backbone()
    vals = []
    for mode in ("graph", "eager", "graph"):
        torch.manual_seed(21)                 # CPU generator: the crop offsets
        torch.cuda.manual_seed(22)            # device generator: probe, rates, centres, noise
        if mode == "graph":
            v = graphed(x, y)
        else:
            bb.zero_grad_flat()
            v = lf(x=x, y=y, model=model)
            v.backward()
        vals.append((float(v.detach()), bb.flat_grads.clone()))
    assert all(np.isfinite(v[0]) for v in vals)
    assert abs(vals[0][0] - vals[2][0]) <= 1e-6 * abs(vals[0][0])
    assert relerr(vals[0][1], vals[2][1]) < 1e-5
    assert abs(vals[0][0] - vals[1][0]) <= 1e-6 * abs(vals[1][0]), (vals[0][0], vals[1][0])
    assert relerr(vals[0][1], vals[1][1]) < 1e-5
    # replays with different seeds give different losses (fresh random numbers reach the graph)
    torch.cuda.manual_seed(23)
    v2 = float(graphed(x, y))
    assert v2 != vals[0][0]
    # no torch RNG kernel was captured: a replay leaves the device generator where it was
    before = torch.cuda.get_rng_state()
    graphed.graph.replay()
    assert torch.equal(before, torch.cuda.get_rng_state())


@pytest.mark.parametrize("B,S,margin", [(32, 48, 6), (4, 48, 0), (7, 20, 3)])
def test_fused_step_draws_are_torch_generator_draws(B, S, margin):
    """sei_proposed_draws (one launch) against the torch calls it replaces -- torch.randn on the probe's interior,
    torch.rand(B), torch.rand(B, 2) + sei_scale_params, torch.randn_like(y) -- from the same generator state: the same
    Philox4x32-10 stream element for element (rates and centres bit for bit; normals bit for bit except where this build's
    logf / sincosf differ from ATen's by an ulp: >= 98 % equal, all within 4e-6), the probe's border zero, and the
    generator left at the same offset, so fused and torch draws may alternate on one random stream."""
    import transforms
    from losses.sure import draw_probe
    y = torch.empty(B, 3, S, S, device="cuda")
    T = transforms.ScalingTransform(kind="padded", antialias=False)
    torch.cuda.manual_seed(1234)
    torch.rand(5, device="cuda")                                   # (a generator that is not at offset 0)
    b_ref = draw_probe(y, margin)
    rate_ref, center_ref = T.sample(B, y.device, y.dtype)
    noise_ref = torch.randn_like(y)
    after_ref = torch.rand(3, device="cuda")

    def same_normals(a, b):
        # (x = s sin(v): an ulp of sincosf / logf is an ABSOLUTE error of ~1e-7 s wherever sin(v) passes through zero)
        return float((a == b).float().mean()) >= 0.98 and torch.allclose(a, b, rtol=4e-6, atol=4e-6)

    args = ref_args()
    import physics
    from losses import get_loss
    lf = get_loss(args, physics.get_physics(args, "cuda")).loss
    lf.sure.margin, lf.sure.cropped_div = margin, True
    assert lf._fused_draws_ok(y)
    torch.cuda.manual_seed(1234)
    torch.rand(5, device="cuda")
    got = lf.draw(y)
    after = torch.rand(3, device="cuda")
    assert same_normals(got["b"], b_ref) and torch.equal(got["b"] == 0, b_ref == 0)
    assert torch.equal(got["rate"], rate_ref) and torch.equal(got["center"], center_ref)
    assert same_normals(got["noise"], noise_ref)
    assert torch.equal(after, after_ref)
    first = {k: v.clone() for k, v in got.items()}
    # ... into static buffers (what a captured step's replay reads): the eager draw's numbers exactly; a second call
    # continues where torch's own calls would
    torch.cuda.manual_seed(1234)
    torch.rand(5, device="cuda")
    assert lf.draw_into(got, y)
    for k in first:
        assert torch.equal(got[k], first[k]), k
    os.environ["SEI_TORCH_DRAWS"] = "1"
    try:
        second_ref = lf.draw(y)                                    # torch's own calls continue the same stream
    finally:
        del os.environ["SEI_TORCH_DRAWS"]
    torch.cuda.manual_seed(1234)
    torch.rand(5, device="cuda")
    lf.draw_into(got, y)
    lf.draw_into(got, y)
    assert torch.equal(got["rate"], second_ref["rate"]) and torch.equal(got["center"], second_ref["center"])
    assert same_normals(got["b"], second_ref["b"]) and same_normals(got["noise"], second_ref["noise"])


def test_crop_pair_on_the_gpu_vs_reference_golden(golden):
    """G14 (src/crop.py run by its own code): the product's crop.CropPair -- forward, and draw_offsets + write_y (the
    sei_crop_window launch a captured step uses) -- on the GPU, same CPU seeds: the same elements in the same places."""
    from crop import CropPair
    g = golden("g14_crop")
    for tag in ("deblur", "deblur_small", "sr2", "sr4", "item3d", "deblur48"):
        xs, ys = tuple(int(v) for v in g[f"{tag}.xshape"]), tuple(int(v) for v in g[f"{tag}.yshape"])
        size, ratio = (int(v) for v in g[f"{tag}.cfg"])
        x = (1 + torch.arange(int(np.prod(xs)), dtype=torch.float32)).view(xs).cuda()
        y = (1 + torch.arange(int(np.prod(ys)), dtype=torch.float32)).view(ys).cuda()
        crop = CropPair("random", size)
        k = 0
        while f"{tag}.seed{k}.xc" in g.files:
            torch.manual_seed(k)
            xc, yc = crop(x, y, xy_size_ratio=ratio)
            assert np.array_equal(xc.cpu().numpy().astype(np.int32), g[f"{tag}.seed{k}.xc"]), (tag, k)
            assert np.array_equal(yc.cpu().numpy().astype(np.int32), g[f"{tag}.seed{k}.yc"]), (tag, k)
            if len(ys) == 4:                                   # the captured step's path: offsets, then one launch
                torch.manual_seed(k)
                i, j, _, _ = crop.draw_offsets(y.shape)
                out = torch.full(tuple(ys[:2]) + (size, size), -1.0, device="cuda")
                crop.write_y(y, i, j, out)
                assert np.array_equal(out.cpu().numpy().astype(np.int32), g[f"{tag}.seed{k}.yc"]), (tag, k)
            k += 1
        xc, yc = CropPair("center", size)(x, y, xy_size_ratio=ratio)
        assert np.array_equal(xc.cpu().numpy().astype(np.int32), g[f"{tag}.center.xc"])
        assert np.array_equal(yc.cpu().numpy().astype(np.int32), g[f"{tag}.center.yc"])


def test_crop_window_kernel_matches_the_padded_crop():
    """sei_crop_window (CropPair.write_y on the GPU) against the reference's pad-then-slice (src/crop.py:26-57 on a 4-D
    batch), windows inside the batch and reaching into the zero rows the batched quirk appends."""
    from crop import CropPair
    crop = CropPair("random", 48)
    y = torch.rand(5, 3, 64, 56, device="cuda")
    for i, j in ((0, 0), (16, 8), (40, 3), (63, 8), (20, 30)):
        out = torch.full((5, 3, 48, 48), 7.0, device="cuda")
        crop.write_y(y, i, j, out)
        padded = torch.nn.functional.pad(y, (0, 48, 0, 48))
        assert torch.equal(out, padded[..., i:i + 48, j:j + 48]), (i, j)


def test_every_captured_step_owns_its_split_k_workspace():
    """The slab workspace of the split-K GEMMs holds per-tile counters: two launches in flight at once must not share one
    (ADVICE r5). A GraphedLossStep records its launches against a workspace of its own, which it keeps alive and which
    leaves the per-stream registry once the graph is built."""
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    steps = []
    for _ in range(2):
        args = bench.reference_args("cuda", 8, 3)
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        steps.append(GraphedLossStep(get_loss(args, p), model, FlatAdam(model, lr=1e-4), (4, 3, 48, 48)))
    a, b = steps[0]._splitk_ws, steps[1]._splitk_ws
    assert a is not None and b is not None and a.data_ptr() != b.data_ptr()
    assert all(ws is not a and ws is not b for ws in _ops._SPLITK_WS.values())
    assert sum(1 for k in _ops._SPLITK_WS if k[0] == 0) <= _ops.SPLITK_WS_STREAMS


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_graphed_training_tracks_eager_training(dtype):
    """Five optimizer steps with the hipGraph-replayed forward+backward against five eager steps from the same
    seeds: same crops, same draws, so per-step losses and gradient norms agree to rounding (float-atomic arrival
    order, amplified over the steps by Adam's sign-like first updates). Regression test for stale split-K partial
    sums inside replayed graphs."""
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype(dtype)
    try:
        runs = {}
        for mode in ("eager", "graph"):
            args = bench.reference_args("cuda", 8, 3)
            torch.manual_seed(0)
            p = physics.get_physics(args, "cuda")
            model = models.get_model(args, p, "cuda").to("cuda")
            bb = model.get_backbone()
            lf = get_loss(args, p)
            opt = FlatAdam(model, lr=1e-4)
            x = torch.rand(8, 3, 256, 256, device="cuda")
            torch.cuda.manual_seed(7)
            y = p(x)
            g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48)) if mode == "graph" else None
            torch.manual_seed(31)
            torch.cuda.manual_seed(32)
            hist = []
            for _ in range(5):
                if g is not None:
                    val = g(x, y)
                else:
                    opt.zero_grad()
                    val = lf(x=x, y=y, model=model)
                    val.backward()
                assert torch.isfinite(bb.flat_grads).all()
                hist.append((float(val.detach()), float(bb.flat_grads.norm())))
                opt.step()
            runs[mode] = hist
        tol_first, tol_last = (1e-6, 2e-3) if dtype == "f32" else (1e-5, 2e-2)
        (le, ge), (lg, gg) = runs["eager"][0], runs["graph"][0]
        assert abs(le - lg) / le < tol_first and abs(ge - gg) / ge < 10 * tol_first, (runs["eager"], runs["graph"])
        for (le, ge), (lg, gg) in zip(runs["eager"], runs["graph"]):
            assert abs(le - lg) / le < tol_last and abs(ge - gg) / ge < tol_last, (runs["eager"], runs["graph"])
        assert runs["graph"][-1][0] < runs["graph"][0][0]          # and the loss goes down
    finally:
        _ops.set_compute_dtype(prev)


def test_optimizer_step_inside_the_weight_gradient_gemm():
    """GraphedLossStep(fuse_optimizer=True): the bottleneck pair's Adam update is applied by the epilogue of the GEMM
    that computes their (merged, complete) gradient instead of storing it. Four steps against the same run with the
    separate optimizer step, same seeds and draws: losses, parameters and both moments agree to the rounding of the
    gradient sum -- the accumulator the epilogue consumes is the value that would have been stored, and the per-element
    arithmetic is sei_adam_fused's. Also: step() without the replay that carries its other half raises."""
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        runs = {}
        for fuse in (False, True):
            args = bench.reference_args("cuda", 8, 3)
            torch.manual_seed(0)
            p = physics.get_physics(args, "cuda")
            model = models.get_model(args, p, "cuda").to("cuda")
            bb = model.get_backbone()
            lf = get_loss(args, p)
            opt = FlatAdam(model, lr=3e-4)
            x = torch.rand(8, 3, 256, 256, device="cuda")
            torch.cuda.manual_seed(7)
            y = p(x)
            g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48), fuse_optimizer=fuse, fuse_min_numel=60000,
                                store_min_numel=0)
            assert len(g.fused_views) == (2 if fuse else 0)
            if fuse:
                assert sorted(tuple(v.shape) for v in g.fused_views) == [(128, 512), (512, 128)]
                covered = sum(hi - lo for lo, hi in opt._step_bounds(bb.flat_params.numel()))
                assert covered == bb.flat_params.numel() - 2 * 65536
            torch.manual_seed(31)
            torch.cuda.manual_seed(32)
            losses = []
            for k in range(4):
                if k == 2:
                    opt.param_groups[0]["lr"] = 1e-4               # a scheduler step between replays
                losses.append(float(g(x, y)))
                opt.step()
            st = opt.state[bb.flat_params]
            runs[fuse] = (losses, bb.flat_params.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(),
                          bb.flat_shadow.clone())
            if fuse:
                with pytest.raises(RuntimeError):
                    opt.step()                                     # no replay carried the fused half of this step
                before = bb.flat_params.clone()
                opt.zero_grad()                                    # an eager step in between (a short last batch)
                lf(x=x, y=y, model=model).backward()
                opt.step()
                lo, hi = opt._fused_ranges[0]
                assert not torch.equal(before[lo:hi], bb.flat_params[lo:hi])       # stepped with the whole bucket
                assert opt.state[bb.flat_params]["step"] == 5
        # (at this size the separate path's storing GEMM splits K over float atomics, so the two runs differ by the
        # rounding of that sum; tests/test_unet_gpu.py::test_weight_gradient_gemm_with_the_adam_epilogue has the equality)
        assert runs[False][0][0] == runs[True][0][0]
        assert all(abs(a - b) < 1e-4 * abs(a) for a, b in zip(runs[False][0], runs[True][0])), (runs[False][0], runs[True][0])
        apart = (runs[True][1] - runs[False][1]).abs()
        # Adam's first steps are sign-like: a weight whose gradient is rounding-small may go the other way in one run
        assert float(apart.mean()) < 2e-6 and float((apart > 3e-5).float().mean()) < 1e-3, (float(apart.mean()), float(apart.max()))
        assert relerr(runs[True][2], runs[False][2]) < 1e-3 and relerr(runs[True][3], runs[False][3]) < 1e-3
        assert runs[True][0][-1] < runs[True][0][0]
    finally:
        _ops.set_compute_dtype(prev)


def test_merged_and_stored_weight_grads_match_plain_accumulation():
    """bf16 mode, default network shape (hidden 32, reduced to 4 scales): the gradients of one proposed-loss
    step are the same whether the two model calls' weight gradients are accumulated one GEMM per call
    (merging off), merged into one two-segment GEMM (eager), or merged AND stored without a prior zero
    (hipGraph replay). Same seeds -> the eager variants see identical random draws."""
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda", 32, 4)
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        x = torch.rand(4, 3, 256, 256, device="cuda")
        y = p(x)
        grads = {}
        for mode in ("plain", "merged"):
            was = _ops.set_weight_grad_merging(mode == "merged", owner=bb)
            try:
                torch.manual_seed(21)
                torch.cuda.manual_seed(22)
                bb.zero_grad_flat()
                records = None
                if mode == "merged":
                    _ops.profile_gemms(True)
                lf(x=x, y=y, model=model).backward()
                if mode == "merged":
                    records = _ops.profile_gemms(False)
                    merged = sum(r[1] == "sei_gemm_bf16nt_dw2" for r in records)          # the merge happened: two-segment
                    for r in records:                                                     # GEMMs, and two-segment jobs of
                        if r[1] == "sei_dwstream_bf16_jobs":                              # the streamed launch (levels 0-1)
                            merged += sum(r[2][0][k].K2 > 0 for k in range(r[2][1]))
                    assert merged > 10, merged
                grads[mode] = bb.flat_grads.clone()
            finally:
                _ops.set_weight_grad_merging(was, owner=bb)
        assert torch.isfinite(grads["merged"]).all()
        # identical bf16 products; the f32 sums over up to 221,184 terms run in another order
        assert relerr(grads["merged"], grads["plain"]) < 2e-4
        # store mode: poison the gradient bucket before each replay -- nothing stale may survive
        g = GraphedLossStep(lf, model, opt, (4, 3, 48, 48))
        assert g.store_weight_grads and bb._sei_zero_ranges is not None
        zeroed = sum(n for _, n in bb._sei_zero_ranges)
        assert 0 < zeroed < 0.05 * bb.flat_grads.numel()              # only the small parameters are zeroed
        outs = []
        for _ in range(2):
            bb.flat_grads.fill_(float("nan"))
            torch.manual_seed(21)
            torch.cuda.manual_seed(22)
            g(x, y)
            outs.append(bb.flat_grads.clone())
        # (two replays of one step differ by the order of the split-K float atomics; a last-bit difference that flips a
        # bf16 rounding downstream -- GEMM operands, and since round 3 the resamplers' -- shows at ~1e-4 of the largest gradient)
        assert torch.isfinite(outs[0]).all() and relerr(outs[0], outs[1]) < 2e-4
        # same seeds -> same crop and same draws as the eager runs: stored == accumulated
        assert relerr(outs[0], grads["merged"]) < 2e-4, relerr(outs[0], grads["merged"])
    finally:
        _ops.set_compute_dtype(prev)


def test_two_models_in_one_process_keep_their_own_step_state():
    """The weight-gradient bookkeeping of models/_ops.py (model calls of the step, parked pairs, merged launches) is per
    backbone: two networks whose steps interleave -- forward A, forward B, backward A, backward B, as when a second
    model is trained or evaluated with gradients beside the first -- each merge their own pairs and end with the
    gradients they get when stepped alone."""
    import bench
    import models
    import physics
    from losses import get_loss
    from models import _ops
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda", 32, 3)
        p = physics.get_physics(args, "cuda")
        lf = get_loss(args, p)
        nets = []
        for seed in (0, 1):
            torch.manual_seed(seed)
            nets.append(models.get_model(args, p, "cuda").to("cuda"))
        bbs = [m.get_backbone() for m in nets]
        assert _ops.state_of(bbs[0]) is not _ops.state_of(bbs[1])
        x = torch.rand(4, 3, 256, 256, device="cuda")
        y = p(x)

        def loss_of(m, seed):
            torch.manual_seed(seed)
            torch.cuda.manual_seed(seed)
            return lf(x=x, y=y, model=m)

        alone = []
        for m, bb in zip(nets, bbs):
            bb.zero_grad_flat()
            loss_of(m, 5).backward()
            alone.append(bb.flat_grads.clone())
            assert len(_ops.merged_weight_grads(bb)) > 4          # the step's two model calls merged, per weight
        for bb in bbs:
            bb.zero_grad_flat()
        la, lb = loss_of(nets[0], 5), loss_of(nets[1], 5)
        assert _ops.state_of(bbs[0])["uses"] == 2 and _ops.state_of(bbs[1])["uses"] == 2
        la.backward()
        lb.backward()
        torch.cuda.synchronize()
        for bb, ref in zip(bbs, alone):
            assert relerr(bb.flat_grads, ref) < 2e-4
            assert not _ops.state_of(bb)["parked"] and len(_ops.merged_weight_grads(bb)) > 4
        assert _ops.state_of()["uses"] == 0                      # nothing leaked into the default state
        # ... and a backbone may carry its own arithmetic mode: an exact-f32 model beside the bf16 one (process default
        # bf16), interleaved with it, gives the gradients it gives alone in an f32 process
        bbs[1].compute_dtype = "f32"
        bbs[1].zero_grad_flat()
        loss_of(nets[1], 5).backward()
        mixed = bbs[1].flat_grads.clone()
        _ops.set_compute_dtype("f32")
        bbs[1].compute_dtype = None
        bbs[1].zero_grad_flat()
        loss_of(nets[1], 5).backward()
        assert relerr(mixed, bbs[1].flat_grads) < 2e-5            # the same exact-f32 launches (split-K atomics aside)
        assert relerr(mixed, alone[1]) > 1e-4                     # and not the bf16 result
    finally:
        _ops.set_compute_dtype(prev)


def test_early_gradient_release_from_inside_the_graph():
    """The captured backward records an event right after the bottleneck block's weight gradients are written
    (an event-record node inserted into the hipGraph); the reducer's side stream waits for it and casts that
    range for the all-reduce while the rest of the backward is still running. Deterministic check on one GPU:
    poison the range, replay, run the reducer's early path, and the bf16 exchange buffer must equal the cast of
    the final gradients everywhere (a wait that did not wait would pick up the poison)."""
    import bench
    import physics
    import models
    import parallel
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda", 32, 4)
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        x = torch.rand(8, 3, 256, 256, device="cuda")
        y = p(x)
        g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48), early_release=True)
        assert g.early_grads is not None
        event, lo, hi = g.early_grads
        assert (hi - lo) > 0.5 * bb.flat_grads.numel()
        red = parallel.FlatGradientReducer(bb.flat_grads, comm_dtype=torch.bfloat16, chunk_mib=16)
        red.set_early_range((lo, hi))
        assert all(not (s < lo < e or s < hi < e) for s, e in red.bounds)          # no chunk straddles the range
        assert sorted(red.order) == list(range(len(red.bounds))) and red._is_early[red.order[0]]
        for _ in range(3):
            bb.flat_grads[lo:hi].fill_(float("nan"))
            red.comm.fill_(float("nan"))
            g(x, y)
            red.reduce_async(early=event)
            torch.cuda.synchronize()
            assert torch.isfinite(red.comm.float()).all()
            assert torch.equal(red.comm, bb.flat_grads.bfloat16())
            opt.step()
    finally:
        _ops.set_compute_dtype(prev)


def test_fused_optimizer_step_on_the_default_network():
    """The 645 M-parameter network, one captured step with and without the optimizer step inside the deep levels'
    weight-gradient GEMMs (same weights, crops and draws): loss, the eight fused weights and their first moments
    agree to the run-to-run noise of the step itself (split-K float atomics in the forward pass), the rest of the
    bucket likewise. At batch 2 the bottleneck reductions (36 + 18 rows) are not a multiple of 8: the
    merged launch does not exist, nothing is fused, and the step still runs."""
    import bench
    import physics
    import models
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda", 32, 5)
        runs = {}
        for fuse in (False, "again", True):           # the plain step twice: the step's own run-to-run noise is the bar
            torch.manual_seed(0)
            p = physics.get_physics(args, "cuda")
            model = models.get_model(args, p, "cuda").to("cuda")
            bb = model.get_backbone()
            lf = get_loss(args, p)
            opt = FlatAdam(model, lr=1e-4)
            x = torch.rand(8, 3, 256, 256, device="cuda")
            torch.cuda.manual_seed(7)
            y = p(x)
            g = GraphedLossStep(lf, model, opt, (8, 3, 48, 48), fuse_optimizer=fuse is True)
            if fuse is True:
                assert len(g.fused_views) == 8, [tuple(v.shape) for v in g.fused_views]
                ranges = list(opt._fused_ranges)          # the bottleneck pair + level 3's two blocks, down and up conv
                assert sum(hi - lo for lo, hi in ranges) == 2 * 8192 * 32768 + 6 * 2048 * 8192      # 98.8 % of the bucket
            torch.manual_seed(31)
            torch.cuda.manual_seed(32)
            loss = float(g(x, y))
            opt.step()
            st = opt.state[bb.flat_params]
            runs[fuse] = (loss, bb.flat_params, st["exp_avg"], st["exp_avg_sq"], bb.flat_shadow) if fuse != "again" else (loss,)
            del g, opt, model, lf
        # Two runs of the SAME step differ: the deep levels' forward GEMMs split K over float atomics, and a last-bit
        # difference there flips bf16 roundings of later activations; at random initialisation the Monte-Carlo divergence
        # (differences of two network outputs over tau = 0.01) carries that into the fourth-fifth digit of the loss (observed
        # between 1e-6 and 1e-4 relative over the boxes of rounds 4-5). The fused step must sit within 3x of what the plain
        # step differs from itself by in this very process (floor 3e-4 = 3x the largest observed); the bit-for-bit statements live at kernel level
        # (tests/test_unet_gpu.py::test_weight_gradient_gemm_with_the_adam_epilogue).
        noise = abs(runs[False][0] - runs["again"][0])
        assert abs(runs[False][0] - runs[True][0]) < max(3e-4 * runs[False][0], 3 * noise), (runs[False][0], runs[True][0], noise)
        for lo, hi in ranges:
            assert relerr(runs[True][2][lo:hi], runs[False][2][lo:hi]) < 2e-2           # exp_avg = 0.1 * gradient
            moved = (runs[True][1][lo:hi] - runs[False][1][lo:hi]).abs()
            assert float(moved.mean()) < 2e-6 and float(moved.max()) <= 2.1e-4          # at most one first Adam step apart
        assert relerr(runs[True][1], runs[False][1]) < 1e-3
        del runs
        torch.cuda.empty_cache()
        # a batch whose bottleneck rows do not merge into one launch: no fusion, same result as the plain step
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        x = torch.rand(2, 3, 256, 256, device="cuda")
        g = GraphedLossStep(lf, model, opt, (2, 3, 48, 48), fuse_optimizer=True)
        assert all(tuple(v.shape) not in ((8192, 32768), (32768, 8192)) for v in g.fused_views)
        assert torch.isfinite(g(x, p(x)))
        opt.step()
    finally:
        _ops.set_compute_dtype(prev)


@pytest.mark.parametrize("graph", [False, True])
def test_joint_backward_of_the_two_model_calls(graph):
    """models/_joint.py: the backward passes of the step's two model calls (2B crops of the SURE term, B measurements of
    the equivariance term: src/losses/__init__.py:133-142 with the default stop_gradient) run as ONE pass over 3B images
    -- arena-allocated activations, the layers' own backward functions played from a tape -- against ordinary autograd over
    the same graph (SEI_NO_JOINT_BACKWARD): same loss, gradients equal up to the float atomics' summation order and the
    bf16 roundings behind it (bar: 3x what two runs of the ordinary path differ by, or 3e-3), and about a third fewer
    launches in the backward pass. Eager and replayed from a hipGraph; a step that differentiates only ONE of the two
    terms falls back to walking that call alone and still matches."""
    import bench
    import models
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _joint, _ops
    from optim import FlatAdam
    import _native as N
    prev = _ops.set_compute_dtype("bf16")
    was = _joint.ENABLED
    try:
        args = bench.reference_args("cuda", 32, 3)
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        B = 4
        gen = torch.Generator().manual_seed(5)
        x = torch.rand((B, 3, 256, 256), generator=gen).cuda()
        y = p(x)
        b = torch.randn((B, 3, 36, 36), generator=gen)
        from losses.sure import embed_probe
        draws = {"b": embed_probe(torch.empty(B, 3, 48, 48, device="cuda"), b.cuda(), 6),
                 "rate": torch.tensor([0.75, 0.5, 0.5, 0.75]).cuda(),
                 "center": (2 * torch.rand((B, 2), generator=gen) - 1).cuda().view(B, 1, 1, 2),
                 "noise": torch.randn((B, 3, 48, 48), generator=gen).cuda()}
        grads, losses, launches = {}, {}, {}
        graphed = {}
        for mode in ("plain", "plain2", "joint"):
            _joint.ENABLED = mode == "joint"
            if graph:
                graphed[mode] = GraphedLossStep(lf, model, opt, (B, 3, 48, 48))
                bb.flat_grads.fill_(float("nan"))
                torch.manual_seed(3)
                val = graphed[mode](x, y, draws=draws)
                launches[mode] = 0
            else:
                opt.zero_grad()
                torch.manual_seed(3)
                N.record_calls(True)
                val = lf(x=x, y=y, model=model, draws=draws)
                nfwd = len(N.record_calls(False))
                N.record_calls(True)
                val.backward()
                launches[mode] = len(N.record_calls(False))
            torch.cuda.synchronize()
            grads[mode], losses[mode] = bb.flat_grads.clone(), float(val)
        assert torch.isfinite(grads["joint"]).all()
        assert abs(losses["joint"] - losses["plain"]) < 1e-5 * abs(losses["plain"])
        noise = relerr(grads["plain2"], grads["plain"])
        assert relerr(grads["joint"], grads["plain"]) < max(3e-3, 3 * noise), (relerr(grads["joint"], grads["plain"]), noise)
        worst = 1.0
        for name, prm in bb.named_parameters():
            off = (prm._sei_grad_view.data_ptr() - bb.flat_grads.data_ptr()) // 4
            a, c = (grads[k][off:off + prm.numel()].double() for k in ("joint", "plain"))
            worst = min(worst, float(a @ c / (a.norm() * c.norm())))
        assert worst > 0.9999, worst
        if not graph:
            assert launches["joint"] < 0.8 * launches["plain"], launches
            # only the equivariance term differentiated: its call is walked alone at the end of the pass
            _joint.ENABLED = True
            opt.zero_grad()
            torch.manual_seed(3)
            lf.loss.keep_outputs = True
            total = lf(x=x, y=y, model=model, draws=draws)
            assert bb._sei_joint.pair.calls == 2 and not bb._sei_joint.pair.broken
            total.backward()
            assert bb._sei_joint.pair.done == [True, True]
            assert relerr(bb.flat_grads, grads["plain"]) < max(3e-3, 3 * noise)
    finally:
        _joint.ENABLED = was
        _ops.set_compute_dtype(prev)


def test_joint_backward_is_not_taken_when_a_torch_add_sits_between_layers():
    """with --ConvolutionalModel__num_conv_blocks > 1 and the default inner residual the encoder stages
    end in a plain torch add `x + xb` (src/models/convolutional.py:226-231) that the tape of layer functions cannot see:
    such a call must stay an ordinary autograd graph (the recorder is marked broken), and its gradients must equal those
    with the joint mechanism switched off -- in particular the gradient that bypasses the blocks through xb."""
    import bench
    import models
    import physics
    from losses import get_loss
    from losses.sure import embed_probe
    from models import _joint, _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    was = _joint.ENABLED
    try:
        args = bench.reference_args("cuda", 32, 3)
        args.ConvolutionalModel__num_conv_blocks = 2
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        bb = model.get_backbone()
        assert len(bb.seq[-1].conv_sequences[0]) == 2 and bb.seq[-1].inner_residual
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        B = 4
        gen = torch.Generator().manual_seed(5)
        x = torch.rand((B, 3, 256, 256), generator=gen).cuda()
        y = p(x)
        draws = {"b": embed_probe(torch.empty(B, 3, 48, 48, device="cuda"), torch.randn((B, 3, 36, 36), generator=gen).cuda(), 6),
                 "rate": torch.tensor([0.75, 0.5, 0.5, 0.75]).cuda(),
                 "center": (2 * torch.rand((B, 2), generator=gen) - 1).cuda().view(B, 1, 1, 2),
                 "noise": torch.randn((B, 3, 48, 48), generator=gen).cuda()}
        grads = {}
        for mode in ("plain", "plain2", "joint"):
            _joint.ENABLED = mode == "joint"
            opt.zero_grad()
            torch.manual_seed(3)
            val = lf(x=x, y=y, model=model, draws=draws)
            if mode == "joint":
                pair = bb._sei_joint.pair
                assert pair.broken and pair.calls == 0          # nothing was cut out of autograd's graph
            val.backward()
            torch.cuda.synchronize()
            grads[mode] = bb.flat_grads.clone()
        noise = relerr(grads["plain2"], grads["plain"])
        assert relerr(grads["joint"], grads["plain"]) < max(1e-3, 3 * noise), (relerr(grads["joint"], grads["plain"]), noise)
        # the first encoder block's gradient carries the bypass: it would be visibly short without it
        w = bb.seq[-1].conv_sequences[0][0].conv1.weight
        off = (w._sei_grad_view.data_ptr() - bb.flat_grads.data_ptr()) // 4
        a, c = (grads[k][off:off + w.numel()].double() for k in ("joint", "plain"))
        assert abs(float(a.norm() / c.norm()) - 1) < 1e-3
    finally:
        _joint.ENABLED = was
        _ops.set_compute_dtype(prev)


def test_second_backward_through_a_walked_call_raises():
    """the joint walk frees the tapes and the arena; a second backward pass through the same graph
    (retain_graph=True) must raise as autograd does, not return without gradients."""
    import bench
    import models
    import physics
    from losses import get_loss
    from models import _joint, _ops
    from optim import FlatAdam
    prev = _ops.set_compute_dtype("bf16")
    try:
        args = bench.reference_args("cuda", 32, 3)
        torch.manual_seed(0)
        p = physics.get_physics(args, "cuda")
        model = models.get_model(args, p, "cuda").to("cuda")
        bb = model.get_backbone()
        lf = get_loss(args, p)
        opt = FlatAdam(model, lr=1e-4)
        x = torch.rand((4, 3, 256, 256)).cuda()
        y = p(x)
        opt.zero_grad()
        val = lf(x=x, y=y, model=model)
        if not (_joint.ENABLED and bb._sei_joint.pair.calls == 2):
            pytest.skip("joint backward not active in this configuration")
        val.backward(retain_graph=True)
        pair = bb._sei_joint.pair
        assert pair.done == [True, True] and pair.bases == [] and pair.base_of == {}     # the 3B-row arena is released
        with pytest.raises(RuntimeError, match="already been walked"):
            val.backward()
    finally:
        _ops.set_compute_dtype(prev)
