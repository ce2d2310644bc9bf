"""Training over a horizon: the bf16 step the bench times (hipGraph replay, stored weight gradients, the optimizer step
inside the weight-gradient GEMM) against float32 training from the same seed, and float32 training against the CPU oracle's
trajectory (reference: demo/train.py:254-273 -- zero_grad, loss, backward, Adam step -- on the proposed deblurring loss,
src/losses/__init__.py:67-142).

No dataset is on disk: the pairs are synthetic images with structure (smooth fields, edges, texture), blurred by the
product's own physics (Gaussian_R2, sigma 5/255), 32 for training and 8 held out.
"""
import math

import numpy as np
import pytest
import torch

from oracle import torch_path as tp

pytestmark = pytest.mark.gpu

HIDDEN, SCALES, BATCH, CROP = 16, 3, 8, 48


def synthetic_images(count, seed, side=256):
    """(count, 3, side, side) in [0, 1]: low-frequency colour fields + a few rectangles and discs + fine texture."""
    g = torch.Generator().manual_seed(seed)
    u = torch.linspace(0, 1, side)
    yy, xx = torch.meshgrid(u, u, indexing="ij")
    out = torch.empty(count, 3, side, side)
    for n in range(count):
        img = torch.zeros(3, side, side)
        for _ in range(6):                                   # smooth fields
            f = torch.rand(2, generator=g) * 6 + 0.5
            ph = torch.rand(2, generator=g) * 2 * math.pi
            amp = torch.rand(3, generator=g)[:, None, None] * 0.25
            img += amp * (torch.sin(2 * math.pi * f[0] * xx + ph[0]) * torch.cos(2 * math.pi * f[1] * yy + ph[1]))[None]
        img += 0.5
        for _ in range(10):                                  # edges
            cx, cy, r = torch.rand(3, generator=g).tolist()
            col = torch.rand(3, generator=g)[:, None, None]
            r = 0.03 + 0.15 * r
            if torch.rand(1, generator=g).item() < 0.5:
                mask = ((xx - cx).abs() < r) & ((yy - cy).abs() < r * (0.3 + torch.rand(1, generator=g).item()))
            else:
                mask = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
            img = torch.where(mask[None], 0.6 * col + 0.4 * img, img)
        f = 20 + 40 * torch.rand(1, generator=g).item()      # texture
        img += 0.06 * torch.sin(2 * math.pi * f * (xx + 0.5 * yy))[None]
        out[n] = img.clamp(0, 1)
    return out


def _setup(dtype, lr):
    import bench
    import models
    import physics
    from losses import get_loss
    from models import _ops
    from optim import FlatAdam
    _ops.set_compute_dtype(dtype)
    args = bench.reference_args("cuda", HIDDEN, SCALES)
    torch.manual_seed(0)
    p = physics.get_physics(args, "cuda")
    model = models.get_model(args, p, "cuda").to("cuda")
    model.train()
    return args, p, model, get_loss(args, p), FlatAdam(model, lr=lr, betas=(0.9, 0.999))


def _psnr_y(model, ys, xs):
    from metrics import psnr_fn
    model.eval()
    with torch.no_grad():
        out = model(ys)
    model.train()
    return float(np.mean([float(psnr_fn(out[i], xs[i])) for i in range(xs.shape[0])]))


def _train(dtype, graphed, steps, train_x, train_y, test_x, test_y, lr=5e-4, evaluate_every=100):
    from graphs import GraphedLossStep
    from models import _ops
    prev = _ops.get_compute_dtype()
    try:
        args, p, model, lf, opt = _setup(dtype, lr)
        g = None
        if graphed:
            g = GraphedLossStep(lf, model, opt, (BATCH, 3, CROP, CROP), fuse_optimizer=True, fuse_min_numel=60000,
                                store_min_numel=0)
            assert g.store_weight_grads and len(g.fused_views) >= 2      # the step the bench times, at this size
        order = torch.Generator().manual_seed(99)
        torch.manual_seed(31)                      # CropPair's host draws
        torch.cuda.manual_seed(32)                 # probe, rates / centres, measurement noise
        losses, psnr = [], {0: _psnr_y(model, test_y, test_x)}
        for k in range(steps):
            idx = torch.randperm(train_x.shape[0], generator=order)[:BATCH].cuda()
            x, y = train_x[idx], train_y[idx]
            if g is not None:
                val = g(x, y)
            else:
                opt.zero_grad()
                val = lf(x=x, y=y, model=model)
                val.backward()
            opt.step()
            losses.append(float(val.detach()))
            if (k + 1) % evaluate_every == 0:
                psnr[k + 1] = _psnr_y(model, test_y, test_x)
        assert all(math.isfinite(v) for v in losses)
        return np.array(losses), psnr
    finally:
        _ops.set_compute_dtype(prev)


def test_bf16_training_tracks_f32_training_over_300_steps():
    """300 optimizer steps, hidden 16 / 3 scales, batch 8, from one seed (same weights, same batches, same crop offsets,
    same device draws): `bf16 + hipGraph + stored weight gradients + Adam in the weight-gradient GEMM` (what bench.py
    times) against eager float32. Bars: held-out PSNR-Y within 0.05 dB at steps 100 / 200 / 300, loss curves within 2 %
    (means over windows of 25 steps; the per-step value carries the Monte-Carlo divergence term and both runs see the
    same draws), and both runs learn (the loss falls, the held-out PSNR rises above the blurred input's)."""
    import physics
    import bench
    args = bench.reference_args("cuda", HIDDEN, SCALES)
    p = physics.get_physics(args, "cuda")
    train_x = synthetic_images(32, 1).cuda()
    test_x = synthetic_images(8, 2, side=96).cuda()
    torch.manual_seed(4321)
    torch.cuda.manual_seed(4321)
    train_y, test_y = p(train_x), p(test_x)
    runs = {}
    for name, dtype, graphed in (("f32", "f32", False), ("bf16", "bf16", True)):
        runs[name] = _train(dtype, graphed, 300, train_x, train_y, test_x, test_y)
    (l32, p32), (l16, p16) = runs["f32"], runs["bf16"]
    report = {"psnr_f32": p32, "psnr_bf16": p16,
              "loss_windows_f32": [float(l32[i:i + 25].mean()) for i in range(0, 300, 25)],
              "loss_windows_bf16": [float(l16[i:i + 25].mean()) for i in range(0, 300, 25)]}
    print("horizon:", report)
    assert abs(l16[0] - l32[0]) < 2e-2 * abs(l32[0]), report           # the first step: one forward/backward in bf16
    for k in (100, 200, 300):
        assert abs(p16[k] - p32[k]) < 0.05, report
    for i in range(0, 300, 25):
        a, b = l32[i:i + 25].mean(), l16[i:i + 25].mean()
        assert abs(a - b) < 0.02 * abs(a), (i, report)
    for losses, psnr in runs.values():
        assert losses[-50:].mean() < losses[:10].mean() and psnr[300] > psnr[0] + 0.3, report


def test_f32_training_follows_the_oracle_trajectory_for_20_steps():
    """20 optimizer steps of the float32 HIP path (losses.ProposedLoss + optim.FlatAdam) against the CPU oracle stepping
    the same loss in float64 with torch.optim.Adam (the reference's optimizer, demo/train.py:186), every random number of
    every step injected into both: per-step loss to 1e-4 relative and the parameter vector after the last step. Adam's
    first updates are sign-like (m / sqrt(v) = +-1), so a weight whose gradient is rounding-small can move the other way
    in one run: the parameter check bounds the mean distance and the fraction of such weights."""
    from losses.sure import embed_probe
    from models import _ops
    prev = _ops.get_compute_dtype()
    try:
        lr = 5e-4
        args, p, model, lf, opt = _setup("f32", lr)
        bb = model.get_backbone()
        sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in model.get_weights().items()}
        ref_opt = torch.optim.Adam(list(sd.values()), lr=lr, betas=(0.9, 0.999))
        kern = tp.blur_kernel("Gaussian_R2")
        A = lambda v: tp.blur_fft(v, kern)
        net = lambda v: tp.unet_forward(sd, v, scales=SCALES)
        gen = torch.Generator().manual_seed(5)
        images = synthetic_images(4, 3, side=64)
        worst = 0.0
        for k in range(20):
            i, j = (int(v) for v in torch.randint(0, 64 - CROP + 1, (2,), generator=gen))
            x = images[:, :, i:i + CROP, j:j + CROP].contiguous()
            y = (tp.blur_fft(x, kern) + 5 / 255 * torch.randn(x.shape, generator=gen)).contiguous()
            b_int = torch.randn((4, 3, CROP - 12, CROP - 12), generator=gen)
            noise = torch.randn((4, 3, CROP, CROP), generator=gen)
            rate = torch.tensor([0.75, 0.5])[torch.randint(0, 2, (4,), generator=gen)]
            center = torch.rand((4, 2), generator=gen) * 2 - 1
            yd = y.cuda()
            draws = {"b": embed_probe(yd, b_int.cuda(), 6), "rate": rate.cuda(), "center": center.cuda().view(4, 1, 1, 2),
                     "noise": noise.cuda()}
            opt.zero_grad()
            val = lf.loss(x=None, y=yd, model=model, draws=draws)
            val.backward()
            opt.step()
            ref_opt.zero_grad()
            ref, _ = tp.proposed_loss(y.double(), A, net, 5 / 255, margin=6, rate=rate.double(),
                                      center=center.double().view(-1, 1, 1, 2), b=b_int.double(), n=noise.double())
            ref.backward()
            ref_opt.step()
            err = abs(float(val.detach()) - float(ref.detach())) / abs(float(ref.detach()))
            worst = max(worst, err)
            assert err < 1e-4, (k, float(val.detach()), float(ref.detach()))
        got = torch.cat([v.detach().reshape(-1) for v in model.get_weights().values()]).cpu().double()
        want = torch.cat([v.detach().reshape(-1) for v in sd.values()])
        apart = (got - want).abs()
        moved = (want - torch.cat([v.reshape(-1) for v in _setup("f32", lr)[2].get_weights().values()]).cpu().double()).abs()
        print(f"trajectory: worst per-step loss rel {worst:.2e}; parameters apart mean {float(apart.mean()):.2e} max "
              f"{float(apart.max()):.2e}; moved mean {float(moved.mean()):.2e}")
        assert float(apart.mean()) < 0.01 * float(moved.mean())
        assert float((apart > 0.1 * 20 * lr).float().mean()) < 1e-3
    finally:
        _ops.set_compute_dtype(prev)
