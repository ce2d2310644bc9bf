"""SwinIR backbone (BASELINE configs[4]) on the GPU against oracle/swinir_path.py.

PARITY UNPINNED: deepinv / timm are absent and the reference holds no SwinIR fixtures, so the oracle is a
restatement of the published architecture (see its header); these tests pin the HIP path to that restatement:
float32 forward <= 1e-4, gradients <= 1e-3 (the bar VERDICT r1 set)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import swinir_path as sp

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _attention_reference(qkv, table, B, H, W, heads, shift):
    """WindowAttention on natural-order tokens by the reference's own steps: roll, window_partition, bias lookup,
    mask, softmax, window_reverse, roll back (float64)."""
    C = qkv.shape[1] // 3
    x = qkv.view(B, H, W, 3 * C)
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = sp.window_partition(x, 8).view(-1, 64, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * (C // heads) ** -0.5, xw[1], xw[2]
    attn = q @ k.transpose(-2, -1)
    idx = sp.relative_position_index(8)
    attn = attn + table[idx.view(-1)].view(64, 64, -1).permute(2, 0, 1).unsqueeze(0)
    if shift:
        mask = sp.shift_mask(H, W, 8, shift).to(attn.dtype)
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, heads, 64, 64) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, 64, 64)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, 8, 8, C)
    out = sp.window_reverse(out, 8, H, W)
    if shift:
        out = torch.roll(out, shifts=(shift, shift), dims=(1, 2))
    return out.reshape(B * H * W, C)


@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("B,H,W,heads,hd", [(2, 16, 16, 6, 30), (1, 24, 16, 3, 32), (3, 48, 48, 6, 30)])
def test_window_attention_fwd_bwd(B, H, W, heads, hd, shift):
    from models import _swin_ops as S
    gen = torch.Generator().manual_seed(B + H + shift)
    C = heads * hd
    qkv = torch.randn((B * H * W, 3 * C), generator=gen)
    table = torch.randn((225, heads), generator=gen) * 0.5
    go = torch.randn((B * H * W, C), generator=gen)
    qd, td = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref = _attention_reference(qd, td, B, H, W, heads, shift)
    rq, rt = torch.autograd.grad(ref, [qd, td], go.double())
    out = S.window_attention(qkv.cuda(), table.cuda(), B, H, W, heads, shift)
    assert relerr(out, ref) < 5e-6
    dtable = torch.zeros_like(table).cuda()
    dqkv = S.window_attention_bwd(qkv.cuda(), table.cuda(), go.cuda(), dtable, B, H, W, heads, shift)
    assert relerr(dqkv, rq) < 2e-5 and relerr(dtable, rt) < 2e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout,act,res", [(2, 16, 16, 180, 180, 0, True), (1, 8, 24, 180, 64, 1, False),
                                                    (2, 12, 12, 64, 256, 0, False)])
def test_conv3x3_on_the_padded_grid(B, H, W, Cin, Cout, act, res):
    from models import _swin_ops as S
    gen = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn((B, H, W, Cin), generator=gen)
    w = torch.randn((Cout, Cin, 3, 3), generator=gen) * (Cin * 9) ** -0.5
    b = torch.randn(Cout, generator=gen)
    r = torch.randn((B, H, W, Cout), generator=gen) if res else None
    go = torch.randn((B, H, W, Cout), generator=gen)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xd.permute(0, 3, 1, 2), wd, bd, padding=1).permute(0, 2, 3, 1)
    if act:
        ref = F.leaky_relu(ref, 0.01)
    if res:
        ref = ref + r.double()
    rx, rw, rb = torch.autograd.grad(ref, [xd, wd, bd], go.double())
    xc = x.cuda().requires_grad_(True)
    wc, bc = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    rc = r.cuda().requires_grad_(True) if res else None
    y = S.Conv3x3GemmFn.apply(xc, wc, bc, rc, act)
    assert relerr(y, ref) < 5e-6
    y.backward(go.cuda())
    assert relerr(xc.grad, rx) < 1e-5 and relerr(wc.grad, rw) < 1e-5 and relerr(bc.grad, rb) < 1e-5
    if res:
        assert torch.equal(rc.grad.cpu(), go)


def _load(model, sd):
    own = model.state_dict()
    model.load_state_dict({k: sd.get(k, own[k]) for k in own})        # buffers (index, mask) keep their own values


@pytest.mark.parametrize("cfg", ["deblur_d22_train", "deblur_full_eval", "sr2_d2_train_ragged", "sr4_d2_eval"])
def test_swinir_model_vs_oracle(cfg):
    from models.swinir import SwinIR
    depths = {"deblur_d22_train": (2, 2), "deblur_full_eval": (6,) * 6, "sr2_d2_train_ragged": (2,), "sr4_d2_eval": (2,)}[cfg]
    up = {"deblur_d22_train": 1, "deblur_full_eval": 1, "sr2_d2_train_ragged": 2, "sr4_d2_eval": 4}[cfg]
    train = "train" in cfg
    shape = (2, 3, 20, 27) if "ragged" in cfg else ((2, 3, 48, 48) if "full" in cfg else (2, 3, 24, 32))
    torch.manual_seed(3)
    model = SwinIR(upscale=up, upsampler="pixelshuffle" if up > 1 else None, depths=depths, num_heads=(6,) * len(depths))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items() if v.dtype.is_floating_point
          and "attn_mask" not in k}
    # make every parameter matter (LayerNorm / bias defaults are 1 / 0)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, v in sd.items():
            if k.endswith("bias") or "norm" in k:
                v.add_(0.1 * torch.randn(v.shape, generator=gen))
    _load(model, {k: v.detach() for k, v in sd.items()})
    model = model.cuda()
    model.train(train)
    x = torch.rand(shape, generator=gen)
    go = torch.randn((shape[0], 3, shape[2] * up, shape[3] * up), generator=gen)
    masks = None
    if train:
        masks = sp.draw_drop_masks(shape[0], depths=depths, generator=gen)
        assert masks[0] is None and masks[-1] is not None
    ref = sp.swinir_forward(sd, x, upscale=up, drop_masks=masks, depths=depths)
    ref.backward(go)
    dev_masks = None if masks is None else [None if m is None else tuple(v.cuda() for v in m) for m in masks]
    model.zero_grad_flat()
    out = model(x.cuda(), drop_masks=dev_masks)
    assert out.shape == ref.shape
    assert relerr(out, ref) < 1e-4, relerr(out, ref)
    out.backward(go.cuda())
    worst = max((relerr(p.grad, sd[k].grad), k) for k, p in model.named_parameters())
    assert worst[0] < 1e-3, worst


def test_swinir_factory_and_training_mode_draws():
    """get_model with the reference's default architecture builds the 11.5 M-parameter SwinIR; training-mode
    forwards draw stochastic-depth masks (two calls differ), eval-mode forwards are deterministic."""
    import bench
    import models
    import physics
    args = bench.reference_args("cuda")
    args.ProposedModel__architecture = "Transformer"
    p = physics.get_physics(args, "cuda")
    torch.manual_seed(0)
    model = models.get_model(args, p, "cuda").to("cuda")
    bb = model.get_backbone()
    assert sum(q.numel() for q in bb.parameters()) == 11_504_163 and bb.flat_params is not None
    assert model.get_parameter("model.model.conv_last.weight").shape == (3, 180, 3, 3)     # demo/train.py:180-184
    y = torch.rand(2, 3, 48, 48, device="cuda")
    model.train()
    torch.cuda.manual_seed(1)
    a, b = model(y), model(y)
    assert not torch.equal(a, b)
    model.eval()
    assert torch.equal(model(y), model(y))


@pytest.mark.parametrize("task", ["deblurring", "sr"])
def test_proposed_loss_step_with_swinir_vs_oracle(task):
    """One proposed-loss step (SURE + scale-EI) around the SwinIR backbone in training mode -- three model calls,
    each with its own stochastic-depth masks, the first two fused into one pass of 2B images -- against the oracle
    with the same masks, probe, rates, centres and noise injected; then the same step replayed from a hipGraph."""
    import bench
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from losses.sure import embed_probe
    from models.swinir import SwinIR
    from optim import FlatAdam
    from oracle import torch_path as tp
    sr = task == "sr"
    up, margin, depths = (2, 0, (2,)) if sr else (1, 6, (2, 2))
    args = bench.reference_args("cuda", task=task, sr_factor=2 if sr else None)
    p = physics.get_physics(args, "cuda")
    lf = get_loss(args, p)
    torch.manual_seed(4)
    model = SwinIR(upscale=up, upsampler="pixelshuffle" if sr else None, depths=depths, num_heads=(6,) * len(depths))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()
          if v.dtype.is_floating_point and "attn_mask" not in k}
    model = model.cuda().train()
    gen = torch.Generator().manual_seed(8)
    B = 2
    y = torch.rand((B, 3, 48, 48), generator=gen)
    b_int = torch.randn((B, 3, 48 - 2 * margin, 48 - 2 * margin), generator=gen)
    noise = torch.randn((B, 3, 48, 48), generator=gen)
    rate, center = torch.tensor([0.5, 0.75]), torch.tensor([[0.2, -0.3], [-0.6, 0.4]])
    masks = [sp.draw_drop_masks(B, depths=depths, rate=0.5, generator=gen) for _ in range(3)]   # (rates of the draw are
    # irrelevant to the arithmetic: any per-sample factors exercise the same path; 0.5 makes zeros likely)
    if sr:
        A = lambda v: tp.downsample_aa(v, 2)
    else:
        k = tp.blur_kernel("Gaussian_R2")
        A = lambda v: tp.blur_fft(v, k)
    calls = iter(masks)
    ref_model = lambda v: sp.swinir_forward(sd, v, upscale=up, drop_masks=next(calls), depths=depths)
    ref, aux = tp.proposed_loss(y, A, ref_model, 5 / 255, margin=margin, rate=rate, center=center.view(B, 1, 1, 2),
                                b=b_int, n=noise)
    ref.backward()

    def dev(m):
        return None if m is None else tuple(v.cuda() for v in m)

    both = [None if a is None else tuple(torch.cat([u, v]).cuda() for u, v in zip(a, b_)) for a, b_ in zip(masks[0], masks[1])]
    yd = y.cuda()
    draws = {"b": embed_probe(yd, b_int.cuda(), margin), "rate": rate.cuda(), "center": center.cuda().view(B, 1, 1, 2),
             "noise": noise.cuda(), "drop": [both, [dev(m) for m in masks[2]]]}
    model.zero_grad_flat()
    val = lf.loss(x=None, y=yd, model=model, draws=draws)
    val.backward()
    assert abs(float(val) - float(ref)) < 1e-4 * abs(float(ref)), (float(val), float(ref))
    worst = max((relerr(q.grad, sd[k_].grad), k_) for k_, q in model.named_parameters())
    assert worst[0] < 2e-3, worst
    # hipGraph replay of the same step through Loss.forward (its crop included): static masks / draws refreshed
    # outside the graph, so replay == eager for equal seeds (crop offsets) and equal injected draws
    x_dummy = torch.zeros((B, 3, 48 * up, 48 * up), device="cuda")
    torch.manual_seed(17)
    model.zero_grad_flat()
    val_e = lf(x=x_dummy, y=yd, model=model, draws=draws)
    val_e.backward()
    eager_grads = model.flat_grads.clone()
    opt = FlatAdam(model, lr=1e-4)
    graphed = GraphedLossStep(lf, model, opt, (B, 3, 48, 48))
    assert graphed.static_draws is not None and "drop" in graphed.static_draws
    model.flat_grads.fill_(float("nan"))
    torch.manual_seed(17)
    val_g = graphed(x_dummy, yd, draws=draws)
    assert abs(float(val_g) - float(val_e)) <= 1e-6 * abs(float(val_e))
    assert relerr(model.flat_grads, eager_grads) < 1e-5
    torch.cuda.manual_seed(3)
    a = float(graphed(x_dummy, yd))
    b2 = float(graphed(x_dummy, yd))                        # fresh masks / draws reach the replay
    assert np.isfinite(a) and a != b2


def test_swinir_timed_configuration_vs_oracle():
    """BASELINE configs[4] as bench.py times it (`--arch swinir --task sr --sr-factor 2`, bf16): the FULL-depth network
    (6 x 6 blocks, 11.75 M parameters, src/models/__init__.py:51-74) in TRAINING mode with injected stochastic-depth
    masks, the throughput path (bf16 LDS-DMA GEMMs on re-laid-out weights, MFMA window attention, implicit-GEMM 3x3
    convolutions), the whole proposed-loss step (x2 antialiased physics, SURE margin 0, scale-EI), replayed from a
    hipGraph -- against the FLOAT64 oracle (oracle/swinir_path.py: PARITY UNPINNED, a restatement of the published
    network; deepinv / timm are absent) on the same weights, masks and draws: restored images within 0.01 dB PSNR-Y, loss
    within bf16 rounding, per-parameter gradient cosines."""
    import bench
    import metrics
    import physics
    from graphs import GraphedLossStep
    from losses import get_loss
    from models import _ops
    from models.swinir import SwinIR
    from optim import FlatAdam
    from oracle import torch_path as tp
    prev = _ops.set_compute_dtype("bf16")
    try:
        depths = (6,) * 6
        args = bench.reference_args("cuda", task="sr", sr_factor=2)
        p = physics.get_physics(args, "cuda")
        lf = get_loss(args, p)
        lf.loss.keep_outputs = True
        torch.manual_seed(4)
        model = SwinIR(upscale=2, upsampler="pixelshuffle", depths=depths, num_heads=(6,) * 6)
        assert sum(q.numel() for q in model.parameters()) == 11_752_487
        gen = torch.Generator().manual_seed(8)
        with torch.no_grad():                       # make every parameter matter (LayerNorm / bias defaults are 1 / 0)
            for k_, v in model.named_parameters():
                if k_.endswith("bias") or "norm" in k_:
                    v.add_(0.05 * torch.randn(v.shape, generator=gen))
        sd = {k_: v.detach().double().requires_grad_(True) for k_, v in model.state_dict().items()
              if v.dtype.is_floating_point and "attn_mask" not in k_}
        model = model.cuda().train()
        B = 2
        x = torch.rand((B, 3, 96, 96), generator=gen)
        y = tp.downsample_aa(x, 2) + 5 / 255 * torch.randn((B, 3, 48, 48), generator=gen)
        b = torch.randn((B, 3, 48, 48), generator=gen)
        noise = torch.randn((B, 3, 48, 48), generator=gen)
        rate, center = torch.tensor([0.5, 0.75]), torch.tensor([[0.2, -0.3], [-0.6, 0.4]])
        masks = [sp.draw_drop_masks(B, depths=depths, generator=gen, dtype=torch.float64) for _ in range(3)]
        assert any(m is not None and float(min(v.min() for v in m)) == 0.0 for mm in masks for m in mm)   # some branch is dropped
        calls = iter(masks)
        ref_model = lambda v: sp.swinir_forward(sd, v, upscale=2, drop_masks=next(calls), depths=depths)
        ref, aux = tp.proposed_loss(y.double(), lambda v: tp.downsample_aa(v, 2), ref_model, 5 / 255, margin=0,
                                    rate=rate.double(), center=center.double().view(B, 1, 1, 2), b=b.double(),
                                    n=noise.double())
        ref.backward()

        def dev(m):
            return None if m is None else tuple(v.float().cuda() for v in m)

        both = [None if a is None else tuple(torch.cat([u, v]).float().cuda() for u, v in zip(a, b_))
                for a, b_ in zip(masks[0], masks[1])]
        draws = {"b": b.cuda(), "rate": rate.cuda(), "center": center.cuda().view(B, 1, 1, 2), "noise": noise.cuda(),
                 "drop": [both, [dev(m) for m in masks[2]]]}
        opt = FlatAdam(model, lr=1e-4)
        lf.crop_fn = None                            # the pair already has the training crop's size (48 / 96)
        graphed = GraphedLossStep(lf, model, opt, (B, 3, 48, 48))
        xd, yd = x.cuda(), y.cuda()
        for _ in range(2):
            model.flat_grads.fill_(float("nan"))
            loss = float(graphed(xd, yd, draws=draws))
        torch.cuda.synchronize()
        x_net = lf.loss.kept["x_net"].float().cpu()
        assert x_net.shape == aux["x_net"].shape == (B, 3, 96, 96)
        for i in range(B):
            d = abs(float(metrics.psnr_fn(x_net[i], x[i])) - float(tp.psnr_y(aux["x_net"][i].detach(), x[i].double())))
            assert d < 0.01, d
        assert relerr(x_net, aux["x_net"]) < 2e-2, relerr(x_net, aux["x_net"])
        assert abs(loss - float(ref)) < 2e-2 * abs(float(ref)), (loss, float(ref))
        assert torch.isfinite(model.flat_grads).all()
        worst_w, worst_o = (1.0, ""), (1.0, "")
        for name, prm in model.named_parameters():
            g, r = prm.grad.double().flatten().cpu(), sd[name].grad.flatten()
            if float(r.norm()) == 0.0:
                assert float(g.norm()) == 0.0, name
                continue
            cos = float(g @ r / (g.norm() * r.norm()))
            assert abs(float(g.norm() / r.norm()) - 1) < 5e-2, (name, float(g.norm()), float(r.norm()))
            if name.endswith("weight") and prm.dim() >= 2:          # the GEMM operands: linear layers and convolutions
                worst_w = min(worst_w, (cos, name))
            else:
                worst_o = min(worst_o, (cos, name))
        print(f"SwinIR sr x2, full depth, train mode, bf16 + hipGraph vs f64 oracle (unpinned): loss {loss:.6f} vs "
              f"{float(ref):.6f}; gradient cosine >= {worst_w[0]:.5f} ({worst_w[1]}) for GEMM weights, >= {worst_o[0]:.5f} "
              f"({worst_o[1]}) for biases / LayerNorm / bias tables")
        assert worst_w[0] > 0.999, worst_w
        assert worst_o[0] > 0.99, worst_o
    finally:
        _ops.set_compute_dtype(prev)


@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("B,H,W,heads", [(2, 16, 16, 6), (1, 24, 16, 3), (5, 48, 48, 6), (12, 48, 48, 6)])
def test_window_attention_mfma_fwd_bwd(B, H, W, heads, shift):
    """The bf16 MFMA window attention (heads padded 30 -> 32, bf16 in / out) against the float64 reference on the
    SAME bf16-rounded inputs: outputs and gradients to bf16 resolution, zero pad dims, bias-table gradient. (An odd head
    count: the last head pair has an idle wave; 12 images: more windows than resident waves -- the backward kernel's
    prefetch of the next item and its two token tables.)"""
    import _native as N
    gen = torch.Generator().manual_seed(B + H + shift + heads)
    M, HP = B * H * W, 32
    q30 = torch.randn((M, 3, heads, 30), generator=gen)
    qkv = torch.zeros((M, 3, heads, HP))
    qkv[..., :30] = q30
    qkv16 = qkv.reshape(M, 3 * heads * HP).bfloat16()
    table = torch.randn((225, heads), generator=gen) * 0.5
    go30 = torch.randn((M, heads, 30), generator=gen)
    go = torch.zeros((M, heads, HP))
    go[..., :30] = go30
    go16 = go.reshape(M, heads * HP).bfloat16()
    # float64 reference on the rounded values, in the unpadded layout
    qd = qkv16.double().view(M, 3, heads, HP)[..., :30].reshape(M, 3 * heads * 30).requires_grad_(True)
    td = table.double().requires_grad_(True)
    ref = _attention_reference(qd, td, B, H, W, heads, shift)
    rq, rt = torch.autograd.grad(ref, [qd, td], go16.double().view(M, heads, HP)[..., :30].reshape(M, heads * 30))
    scale = 30 ** -0.5
    out16 = torch.full((M, heads * HP), float("nan"), dtype=torch.bfloat16, device="cuda")
    qc, tc, gc = qkv16.cuda(), table.cuda(), go16.cuda()
    lse = torch.full((heads, M), float("nan"), dtype=torch.float32, device="cuda")
    N.call("sei_swin_attn_fwd_bf16", qc.data_ptr(), tc.data_ptr(), out16.data_ptr(), lse.data_ptr(), B, H, W, heads, shift,
           scale)
    assert bool(torch.isfinite(lse).all())
    out = out16.float().cpu().view(M, heads, HP)
    assert float(out[..., 30:].abs().max()) == 0.0
    assert relerr(out[..., :30].reshape(M, -1), ref) < 1e-2
    dqkv16 = torch.full((M, 3 * heads * HP), float("nan"), dtype=torch.bfloat16, device="cuda")
    dtable = torch.zeros_like(tc)
    N.call("sei_swin_attn_bwd_bf16", qc.data_ptr(), tc.data_ptr(), out16.data_ptr(), lse.data_ptr(), gc.data_ptr(),
           dqkv16.data_ptr(), dtable.data_ptr(), B, H, W, heads, shift, scale)
    dq = dqkv16.float().cpu().view(M, 3, heads, HP)
    assert float(dq[..., 30:].abs().max()) == 0.0
    assert relerr(dq[..., :30].reshape(M, -1), rq) < 2e-2
    assert relerr(dtable, rt) < 2e-2
    # run-to-run: the outputs involve no atomics
    out_b = torch.empty_like(out16)
    N.call("sei_swin_attn_fwd_bf16", qc.data_ptr(), tc.data_ptr(), out_b.data_ptr(), 0, B, H, W, heads, shift, scale)
    assert torch.equal(out16, out_b)


@pytest.mark.parametrize("cfg", ["deblur_train", "sr2_eval"])
def test_swinir_bf16_path_tracks_f32(cfg):
    """Throughput mode (bf16 LDS-DMA GEMMs on re-laid-out weights, MFMA window attention, implicit-GEMM 3x3
    convolutions) against the exact-f32 HIP path on the same weights, inputs and stochastic-depth masks: restored
    images to bf16 resolution, every parameter gradient aligned (cosine), and the staged weight gradients land in the
    right places of the flat bucket (a wrong index map shows as a cosine near 0)."""
    from models import _ops
    from models.swinir import SwinIR
    up = 2 if cfg.startswith("sr2") else 1
    depths = (2, 2)
    torch.manual_seed(6)
    model = SwinIR(upscale=up, upsampler="pixelshuffle" if up > 1 else None, depths=depths, num_heads=(6, 6)).cuda()
    gen = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.endswith("bias") or "norm" in k:
                v.add_(0.1 * torch.randn(v.shape, generator=gen).cuda())
    model.train(cfg.endswith("train"))
    x = torch.rand((2, 3, 32, 40), generator=gen).cuda()
    go = torch.randn((2, 3, 32 * up, 40 * up), generator=gen).cuda()
    masks = None
    if model.training:
        masks = [None if m is None else tuple(v.cuda() for v in m)
                 for m in sp.draw_drop_masks(2, depths=depths, rate=0.4, generator=gen)]
    outs = {}
    for mode in ("f32", "bf16"):
        prev = _ops.set_compute_dtype(mode)
        try:
            model.zero_grad_flat()
            out = model(x, drop_masks=masks)
            out.backward(go)
            torch.cuda.synchronize()
            outs[mode] = (out.detach().clone(), model.flat_grads.clone())
        finally:
            _ops.set_compute_dtype(prev)
    assert relerr(outs["bf16"][0], outs["f32"][0]) < 3e-2, relerr(outs["bf16"][0], outs["f32"][0])
    base = model.flat_params.data_ptr()
    worst = (1.0, "")
    for k, p in model.named_parameters():
        off = (p.data_ptr() - base) // 4
        a, b = outs["bf16"][1][off:off + p.numel()].double(), outs["f32"][1][off:off + p.numel()].double()
        if float(b.norm()) == 0.0:                      # a branch dropped for every sample of the batch
            assert float(a.norm()) == 0.0, k
            continue
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        worst = min(worst, (cos, k))
        assert 0.5 < float(a.norm() / (b.norm() + 1e-300)) < 2.0, (k, float(a.norm()), float(b.norm()))
    assert worst[0] > 0.98, worst


@pytest.mark.parametrize("extra", [[], ["--compute_dtype", "bf16"], ["--task", "sr", "--sr_factor", "2", "--compute_dtype", "bf16"]])
def test_train_script_with_the_reference_default_architecture(tmp_path, extra):
    """`train.py` with the reference's DEFAULT --ProposedModel__architecture (Transformer = SwinIR): proposed loss,
    hipGraph replay with static stochastic-depth masks, fused Adam, checkpoint with the published key layout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "run"
    cmd = [sys.executable, os.path.join(root, "train.py"), "--device", "cuda", "--method", "proposed", "--task",
           "deblurring", "--kernel", "Gaussian_R2", "--dataset", "synthetic", "--batch_size", "2", "--epochs", "4",
           "--max_steps", "2", "--out_dir", str(out)] + extra
    env = dict(os.environ, SEI_TRACE_STEP_KIND="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "step kind: hipGraph replay" in r.stdout
    rows = open(out / "training.csv").read().strip().splitlines()
    assert len(rows) == 5 and all(np.isfinite(float(v.split(",")[1])) for v in rows[1:])
    w = torch.load(out / "weights.pt", map_location="cpu")
    assert "conv_last.weight" in w and "layers.5.residual_group.blocks.5.attn.relative_position_bias_table" in w
    assert "layers.0.residual_group.blocks.1.attn_mask" in w          # buffers travel, as in the published weights


def test_conv_weight_gradient_taps_in_one_launch():
    """sei_gemm_bf16nt_dw2_taps == nine accumulating sei_gemm_bf16nt_dw2 launches on the row-shifted grid (two K
    segments = the step's two model calls, and one), to the rounding of the split-K sums."""
    import ctypes
    import _native as N
    Wp, C = 18, 192
    offs = [(ky - 1) * Wp + (kx - 1) for ky in range(3) for kx in range(3)]
    rows_c = (ctypes.c_int * 9)(*offs)
    guard = Wp + 9
    gen = torch.Generator(device="cuda").manual_seed(8)

    def grid(count):
        R8 = (count * Wp * Wp + 7) // 8 * 8
        return (R8, (0.1 * torch.randn((R8, C), device="cuda", generator=gen)).bfloat16(),
                torch.randn((R8 + 2 * guard, C), device="cuda", generator=gen).bfloat16())

    K1, g1, x1 = grid(6)
    K2, g2, x2 = grid(3)
    for two in (True, False):
        ref = torch.zeros((9, C, C), device="cuda")
        out = torch.zeros((9, C, C), device="cuda")
        for t, o in enumerate(offs):
            if two:
                N.call("sei_gemm_bf16nt_dw2", g1.data_ptr(), g2.data_ptr(), C, x1[guard + o:].data_ptr(),
                       x2[guard + o:].data_ptr(), C, ref[t].data_ptr(), C, C, K1, K2, 1)
            else:
                N.call("sei_gemm_bf16nt_ex", g1.data_ptr(), C, 1, x1[guard + o:].data_ptr(), C, 1, ref[t].data_ptr(), None,
                       C, C, K1, 5, None, None, None, None, 0, 0)
        N.call("sei_gemm_bf16nt_dw2_taps", g1.data_ptr(), g2.data_ptr() if two else g1.data_ptr(), C, x1[guard:].data_ptr(),
               x2[guard:].data_ptr() if two else x1[guard:].data_ptr(), C, out.data_ptr(), C, C, K1, K2 if two else 0, 1, 9,
               ctypes.cast(rows_c, ctypes.c_void_p), C * C)
        assert relerr(out, ref) < 1e-5, relerr(out, ref)
        dense = torch.einsum("rm,rn->mn", g1.float(), x1[guard + offs[5]:guard + offs[5] + K1].float())
        if not two:
            assert relerr(out[5], dense) < 1e-4


@pytest.mark.parametrize("K1,K2", [(4608, 2304), (64, 0), (320, 192), (50 * 64, 0), (24576, 12288), (66 * 64 * 32, 0)])
def test_token_streamed_weight_gradients(K1, K2):
    """sei_tokgrad_bf16 / sei_tokgrad_bf16_blocks (nn.Linear's weight gradient dY^T X of deepinv's SwinIR blocks, both
    operands token-major, the step's two model calls as two segments) against the float32 product of the same bf16
    operands, on top of a running gradient: one launch per weight, and the four weights of a block (3 + 1 + 2 + 2
    192 x 192 blocks, operands with padding columns beyond the block) in one launch. Token counts from one 64-token
    k-tile (most workgroups idle) to more k-tiles than workgroups; the last two give every workgroup of the eight-block
    launch (the last one: of every launch) eight stages or more -- the kernel whose stages travel through registers,
    with the segment boundary inside a workgroup's range and stage counts of every remainder modulo 3."""
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(K1 + K2)
    blocks, expect = [], []
    for Mo, Ni, ldy, ldx in [(576, 192, 576, 192), (192, 192, 200, 192), (384, 192, 384, 256), (192, 384, 192, 384)]:
        ys = [(0.5 * torch.randn((k, ldy), device="cuda", generator=gen)).bfloat16() for k in (K1, K2) if k]
        xs = [torch.randn((k, ldx), device="cuda", generator=gen).bfloat16() for k in (K1, K2) if k]
        base = torch.randn((Mo, Ni), device="cuda", generator=gen)
        ref = base + sum(y[:, :Mo].float().T @ x[:, :Ni].float() for y, x in zip(ys, xs))
        d = base.clone()
        assert N.lib().sei_tokgrad_bf16_eligible(Mo, Ni, ldy, ldx, K1, K2) == (Mo // 192) * (Ni // 192)
        N.call("sei_tokgrad_bf16", ys[0].data_ptr(), ys[-1].data_ptr(), ldy, xs[0].data_ptr(), xs[-1].data_ptr(), ldx,
               d.data_ptr(), Ni, Mo, Ni, K1, K2)
        assert relerr(d, ref) < 2e-5, (Mo, Ni, relerr(d, ref))
        dg = base.clone()
        for gy in range(Mo // 192):
            for gx in range(Ni // 192):
                blocks.append(N.TokGradBlock(ys[0].data_ptr(), ys[-1].data_ptr(), xs[0].data_ptr(), xs[-1].data_ptr(), ldy, ldx,
                                             192 * gy, 192 * gx, dg.data_ptr() + 4 * (192 * gy * Ni + 192 * gx), Ni))
        expect.append((dg, ref, ys, xs))
    arr = (N.TokGradBlock * len(blocks))(*blocks)
    N.call("sei_tokgrad_bf16_blocks", arr, len(blocks), K1, K2)
    for dg, ref, _, _ in expect:
        assert relerr(dg, ref) < 2e-5, relerr(dg, ref)
    # shapes the kernel does not take are refused, not mangled
    assert N.lib().sei_tokgrad_bf16_eligible(180, 192, 192, 192, 64, 0) == 0
    assert N.lib().sei_tokgrad_bf16_eligible(192, 192, 192, 192, 72, 0) == 0
    assert N.lib().sei_tokgrad_bf16_eligible(576, 576, 576, 576, 64, 0) == 0


_ROWGEMM_CASES = [("qkv", 576, 192, 576, 1, True), ("proj", 192, 192, 180, 3, False), ("proj_drop", 192, 192, 180, 7, False),
                  ("fc1", 384, 192, 384, 2, False), ("fc2", 192, 384, 180, 3, False), ("fc2_drop", 192, 384, 180, 7, False),
                  ("fc2_dgrad", 384, 192, 384, 4, True), ("fc1_dgrad", 192, 384, 192, 0, False),
                  ("proj_dgrad", 192, 192, 192, 0, True), ("qkv_dgrad", 192, 576, 192, 0, False)]


@pytest.mark.parametrize("M", [64, 4608, 256 * 64 + 128])
@pytest.mark.parametrize("name,Nn,K,nv,epi,only16", _ROWGEMM_CASES, ids=[c[0] for c in _ROWGEMM_CASES])
def test_row_streaming_linear_layers(M, name, Nn, K, nv, epi, only16):
    """sei_rowgemm_bf16 (nn.Linear forward and data gradient of deepinv's SwinIR blocks with the layer's matrix held in
    registers) against float64 on the same bf16 operands, every epilogue the blocks use: float32 outputs to 2e-6 of the
    largest value, bf16 outputs to one rounding; padding columns of bf16 outputs are exact zeros; the float32 output is
    not touched past its nv columns. One tile, fewer tiles than workgroups, more tiles than workgroups (ragged shares).
    The tiled kernel (sei_gemm_bf16nt) on the same inputs agrees to the float summation order."""
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(M + Nn + K + epi)
    a = torch.randn((M, K), device="cuda", generator=gen).bfloat16()
    w = (0.1 * torch.randn((Nn, K), device="cuda", generator=gen)).bfloat16()
    w[nv:] = 0
    bias = torch.randn(nv, device="cuda", generator=gen)
    rows = torch.randn((M, nv), device="cuda", generator=gen)
    drop = (torch.rand(M, device="cuda", generator=gen) > 0.2).float() / 0.8
    assert N.lib().sei_rowgemm_bf16_eligible(M, Nn, K, epi, int(only16)) == 1
    ld32 = nv + 4                                          # a guard column block: must stay untouched
    d32 = None if only16 else torch.full((M, ld32), 7.0, device="cuda")
    d16 = torch.full((M, Nn), 7.0, device="cuda").bfloat16() if (only16 or epi == 2) else None
    R1 = drop if epi == 7 else (rows if epi in (3, 4) else None)
    R2 = rows if epi == 7 else None
    N.call("sei_rowgemm_bf16", a.data_ptr(), K, w.data_ptr(), K, N.ptr(d32), ld32, N.ptr(d16), Nn, M, Nn, K, nv, epi,
           bias.data_ptr() if epi in (1, 2, 3, 7) else None, N.ptr(R1), N.ptr(R2), nv)
    acc = a.double() @ w.double().T
    if epi in (1, 2, 3, 7):
        acc[:, :nv] += bias.double()
    if epi == 3:
        ref = acc[:, :nv] + rows.double()
    elif epi == 7:
        ref = rows.double() + drop.double()[:, None] * acc[:, :nv]
    elif epi == 4:
        x = rows.double()
        ref = acc[:, :nv] * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5)
    else:
        ref = acc[:, :nv]
    scale = float(ref.abs().max())
    if d32 is not None:
        assert float((d32[:, :nv].double() - ref).abs().max()) < 2e-6 * scale
        assert bool((d32[:, nv:] == 7.0).all())
    if d16 is not None:
        r16 = torch.nn.functional.gelu(ref) if epi == 2 else ref
        assert float((d16[:, :nv].double() - r16).abs().max()) < 2.0 ** -8 * max(float(r16.abs().max()), 1.0)
        if nv < Nn:
            assert bool((d16[:, nv:] == 0).all())
    if epi != 2 and not (only16 and nv != Nn):             # the tiled kernel on the same inputs
        o32 = None if only16 else torch.zeros((M, nv), device="cuda")
        o16 = torch.zeros((M, nv), device="cuda").bfloat16() if only16 else None
        N.call("sei_gemm_bf16nt", a.data_ptr(), K, 0, w.data_ptr(), K, 0, N.ptr(o32), N.ptr(o16), M, nv, K, epi,
               bias.data_ptr() if epi in (1, 3, 7) else None, N.ptr(R1), N.ptr(R2), None)
        if o32 is not None:
            assert float((d32[:, :nv] - o32).abs().max()) < 2e-6 * scale
        else:
            assert float((d16[:, :nv].float() - o16.float()).abs().max()) <= 2.0 ** -7 * max(scale, 1.0)


def test_row_streaming_kernel_refuses_other_shapes():
    import _native as N
    elig = N.lib().sei_rowgemm_bf16_eligible
    assert elig(4608, 576, 192, 1, 1) == 1
    assert elig(4600, 576, 192, 1, 1) == 0                 # rows: whole 64-row tiles only
    assert elig(4608, 540, 192, 1, 1) == 0 and elig(4608, 576, 180, 1, 1) == 0
    assert elig(4608, 576, 192, 1, 0) == 0                 # qkv is built for bf16 output only
    assert elig(4608, 192, 576, 3, 0) == 0                 # no residual epilogue at K = 576
    a = torch.zeros((64, 192), device="cuda").bfloat16()
    with pytest.raises(N.NativeLibraryError):
        N.call("sei_rowgemm_bf16", a.data_ptr(), 192, a.data_ptr(), 192, None, 0, a.data_ptr(), 576, 64, 576, 192, 576, 1,
               None, None, None, 0)                        # SEI_EPI_BIAS without a bias


@pytest.mark.parametrize("M,K,cast", [(64, 384, True), (4608, 384, True), (4608, 384, False), (4608, 576, False),
                                      (300 * 64, 576, False), (300 * 64, 384, True)])
def test_layernorm_backward_inside_the_data_gradient(M, K, cast):
    """sei_rowgemm_lnbwd_bf16: nn.Linear's data gradient + nn.LayerNorm's backward + the residual gradient (+ the bf16
    cast / stochastic-depth scale / column sums that feed the next weight gradient) in one launch, against float64
    autograd of the same composition on the same bf16 operands."""
    import _native as N
    C, CP = 180, 192
    gen = torch.Generator(device="cuda").manual_seed(M + K + int(cast))
    a = torch.randn((M, K), device="cuda", generator=gen).bfloat16()
    w = (0.1 * torch.randn((CP, K), device="cuda", generator=gen)).bfloat16()
    w[C:] = 0
    x = torch.randn((M, C), device="cuda", generator=gen) * 2 + 0.3
    gamma = torch.randn(C, device="cuda", generator=gen)
    res = torch.randn((M, C), device="cuda", generator=gen)
    drop = (torch.rand(M, device="cuda", generator=gen) > 0.2).float() / 0.8
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    gg0, gb0, cs0 = (torch.randn(C, device="cuda", generator=gen) for _ in range(3))
    gg, gb, cs = gg0.clone(), gb0.clone(), cs0.clone()
    gx = torch.full((M, C), 7.0, device="cuda")
    y16 = torch.full((M, CP), 7.0, device="cuda").bfloat16() if cast else None
    work = torch.empty(N.lib().sei_rowgemm_lnbwd_work_floats(C), device="cuda")
    assert N.lib().sei_rowgemm_lnbwd_bf16_eligible(M, K, C) == 1
    N.call("sei_rowgemm_lnbwd_bf16", a.data_ptr(), K, w.data_ptr(), K, M, K, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), res.data_ptr(), gx.data_ptr(), C, gg.data_ptr(), gb.data_ptr(), drop.data_ptr() if cast else None,
           N.ptr(y16), CP, cs.data_ptr() if cast else None, work.data_ptr(), work.numel())
    # float64 reference through autograd: y = LayerNorm(x) (weight gamma), loss = sum(y * gh) -> dx, dgamma; dbeta = sum gh
    gh = (a.double() @ w.double().T)[:, :C]
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (C,), gd, torch.zeros(C, device="cuda", dtype=torch.float64), 1e-5)
    dx, dg = torch.autograd.grad((y * gh).sum(), (xd, gd))
    ref = dx + res.double()
    assert float((gx.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert float((gg.double() - gg0.double() - dg).abs().max()) < 2e-5 * float(dg.abs().max())
    assert float((gb.double() - gb0.double() - gh.sum(0)).abs().max()) < 2e-5 * float(gh.sum(0).abs().max()) + 1e-4
    if cast:
        ys = ref * drop.double()[:, None]
        assert float((y16[:, :C].double() - ys).abs().max()) < 2.0 ** -8 * float(ys.abs().max())
        assert bool((y16[:, C:] == 0).all())
        assert float((cs.double() - cs0.double() - ys.sum(0)).abs().max()) < 2e-5 * float(ys.abs().sum(0).max())
    with pytest.raises(N.NativeLibraryError):              # the bf16 copy exists next to K = 384 only
        N.call("sei_rowgemm_lnbwd_bf16", a.data_ptr(), K, w.data_ptr(), K, M, 576, x.data_ptr(), gamma.data_ptr(),
               mean.data_ptr(), rstd.data_ptr(), res.data_ptr(), gx.data_ptr(), C, gg.data_ptr(), gb.data_ptr(),
               drop.data_ptr(), work.data_ptr(), CP, cs.data_ptr(), work.data_ptr(), work.numel())


@pytest.mark.parametrize("M", [64, 4608, 256 * 64 + 192])
def test_gelu_gradient_with_the_recomputed_preactivation(M):
    """sei_rowgemm_dgelu_bf16 recomputes fc1's pre-activation from fc1's input instead of reading the stored one: the same
    bits as sei_rowgemm_bf16(SEI_EPI_MUL_DGELU) fed with the float32 pre-activation the forward kernel writes, and the
    forward kernel's bf16 gelu output does not depend on whether that float32 output is requested."""
    import _native as N
    Nn, K, nv = 384, 192, 360
    gen = torch.Generator(device="cuda").manual_seed(M)
    h2 = torch.randn((M, K), device="cuda", generator=gen).bfloat16()
    w1 = (0.1 * torch.randn((Nn, K), device="cuda", generator=gen)).bfloat16()
    w1[nv:] = 0
    b1 = torch.zeros(Nn, device="cuda")
    b1[:nv] = torch.randn(nv, device="cuda", generator=gen)
    gy = torch.randn((M, K), device="cuda", generator=gen).bfloat16()
    w2t = (0.1 * torch.randn((Nn, K), device="cuda", generator=gen)).bfloat16()
    w2t[nv:] = 0
    f3 = torch.empty((M, Nn), device="cuda")
    f4, f4b = (torch.empty((M, Nn), device="cuda").bfloat16() for _ in range(2))
    N.call("sei_rowgemm_bf16", h2.data_ptr(), K, w1.data_ptr(), K, f3.data_ptr(), Nn, f4.data_ptr(), Nn, M, Nn, K, Nn, 2,
           b1.data_ptr(), None, None, 0)
    N.call("sei_rowgemm_bf16", h2.data_ptr(), K, w1.data_ptr(), K, None, 0, f4b.data_ptr(), Nn, M, Nn, K, Nn, 2,
           b1.data_ptr(), None, None, 0)
    assert torch.equal(f4, f4b)
    stored, recomputed = (torch.full((M, Nn), 7.0, device="cuda").bfloat16() for _ in range(2))
    N.call("sei_rowgemm_bf16", gy.data_ptr(), K, w2t.data_ptr(), K, None, 0, stored.data_ptr(), Nn, M, Nn, K, Nn, 4, None,
           f3.data_ptr(), None, Nn)
    assert N.lib().sei_rowgemm_dgelu_bf16_eligible(M, Nn, K) == 1
    N.call("sei_rowgemm_dgelu_bf16", gy.data_ptr(), K, w2t.data_ptr(), K, h2.data_ptr(), K, w1.data_ptr(), K, b1.data_ptr(), Nn,
           recomputed.data_ptr(), Nn, M, Nn, K)
    assert torch.equal(stored, recomputed)
    x = (h2.double() @ w1.double().T + b1.double())
    ref = (gy.double() @ w2t.double().T) * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5)
    assert float((recomputed.double() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max())
    assert bool((recomputed[:, nv:] == 0).all())


@pytest.mark.parametrize("M,K,drop", [(64, 192, True), (4608, 192, False), (4608, 384, True), (300 * 64, 384, False),
                                      (300 * 64, 192, True)])
def test_residual_linear_layer_with_the_layernorm_behind_it(M, K, drop):
    """sei_rowgemm_ln_bf16 (proj -> norm2, fc2 -> the next block's norm1 of deepinv's SwinIR) against the two launches it
    replaces: the residual rows agree with sei_rowgemm_bf16 bit for bit, mean / rstd / the bf16 LayerNorm rows with
    sei_ln_fwd_bf16_pad to the last bits (sums over 16 lanes instead of 64), and both with float64."""
    import _native as N
    C, CP = 180, 192
    gen = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn((M, K), device="cuda", generator=gen).bfloat16()
    w = (0.1 * torch.randn((CP, K), device="cuda", generator=gen)).bfloat16()
    w[C:] = 0
    bias, gamma, beta = (torch.randn(C, device="cuda", generator=gen) for _ in range(3))
    res = torch.randn((M, C), device="cuda", generator=gen) * 2 + 0.5
    scale = (torch.rand(M, device="cuda", generator=gen) > 0.2).float() / 0.8 if drop else None
    out = torch.full((M, C), 7.0, device="cuda")
    h = torch.full((M, CP), 7.0, device="cuda").bfloat16()
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    assert N.lib().sei_rowgemm_ln_bf16_eligible(M, K, C) == 1
    N.call("sei_rowgemm_ln_bf16", a.data_ptr(), K, w.data_ptr(), K, M, K, C, bias.data_ptr(), N.ptr(scale), res.data_ptr(),
           out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-5, 1, h.data_ptr(), CP, mean.data_ptr(), rstd.data_ptr())
    out2 = torch.empty((M, C), device="cuda")
    N.call("sei_rowgemm_bf16", a.data_ptr(), K, w.data_ptr(), K, out2.data_ptr(), C, None, 0, M, CP, K, C, 7 if drop else 3,
           bias.data_ptr(), scale.data_ptr() if drop else res.data_ptr(), res.data_ptr() if drop else None, C)
    assert torch.equal(out, out2)
    h2 = torch.empty((M, CP), device="cuda").bfloat16()
    mean2, rstd2 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    N.call("sei_ln_fwd_bf16_pad", out2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), h2.data_ptr(), mean2.data_ptr(),
           rstd2.data_ptr(), M, C, CP, 1e-5, 1)
    assert float((mean - mean2).abs().max()) < 1e-6 * float(mean2.abs().max())
    assert float((rstd - rstd2).abs().max()) < 2e-6 * float(rstd2.abs().max())
    assert float((h.float() - h2.float()).abs().max()) <= 2.0 ** -7 * float(h2.float().abs().max())
    assert bool((h[:, C] == 1).all()) and bool((h[:, C + 1:] == 0).all())
    ref = torch.nn.functional.layer_norm(out.double(), (C,), gamma.double(), beta.double(), 1e-5)
    assert float((h[:, :C].double() - ref).abs().max()) < 2.0 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize("B,H,W,Cin,Cout,act,with_res", [(2, 16, 24, 192, 180, 0, True), (3, 48, 48, 192, 180, 0, False),
                                                          (1, 24, 16, 64, 4, 0, False), (2, 16, 16, 192, 64, 1, False),
                                                          (2, 8, 8, 192, 180, 1, True)])
def test_conv_gemm_with_the_unpad_epilogue(B, H, W, Cin, Cout, act, with_res):
    """sei_gemm_bf16nt_conv_unpad == sei_gemm_bf16nt_conv followed by sei_unpad_nhwc (border pixels dropped, LeakyReLU,
    residual) to the float summation order, and the dense convolution of the same bf16 operands."""
    import ctypes
    import _native as N
    gen = torch.Generator(device="cuda").manual_seed(B * H + W + Cout)
    Wp, R = W + 2, B * (H + 2) * (W + 2)
    guard = Wp + 9
    x = torch.randn((B, H, W, Cin), device="cuda", generator=gen)
    xp = torch.empty((R + 2 * guard, Cin), device="cuda").bfloat16()
    N.call("sei_pad_nhwc_bf16", x.data_ptr(), xp.data_ptr(), B, H, W, Cin, Cin, guard)
    w = (0.05 * torch.randn((Cout, 9 * Cin), device="cuda", generator=gen)).bfloat16()
    bias = torch.randn(Cout, device="cuda", generator=gen)
    res = torch.randn((B, H, W, Cout), device="cuda", generator=gen) if with_res else None
    offs = (ctypes.c_int * 9)(*[(ky - 1) * Wp + (kx - 1) for ky in range(3) for kx in range(3)])
    outp = torch.empty((R, Cout), device="cuda")
    N.call("sei_gemm_bf16nt_conv", xp[guard:].data_ptr(), Cin, offs, w.data_ptr(), 9 * Cin, outp.data_ptr(), None, R, Cout, 1,
           bias.data_ptr())
    ref = torch.empty((B, H, W, Cout), device="cuda")
    N.call("sei_unpad_nhwc", outp.data_ptr(), N.ptr(res), ref.data_ptr(), B, H, W, Cout, act)
    y = torch.full((B, H, W, Cout), 7.0, device="cuda")
    N.call("sei_gemm_bf16nt_conv_unpad", xp[guard:].data_ptr(), Cin, offs, w.data_ptr(), 9 * Cin, y.data_ptr(), N.ptr(res), B, H,
           W, Cout, bias.data_ptr(), act)
    assert relerr(y, ref) < 2e-6                           # (the unfused GEMM may split K on small grids: another summation order)
    dense = torch.nn.functional.conv2d(x.bfloat16().float().permute(0, 3, 1, 2),
                                       w.float().view(Cout, 9, Cin).permute(0, 2, 1).reshape(Cout, Cin, 3, 3), bias, padding=1)
    if act:
        dense = torch.nn.functional.leaky_relu(dense, 0.01)
    dense = dense.permute(0, 2, 3, 1) + (res if with_res else 0)
    assert relerr(y, dense) < 1e-4


@pytest.mark.parametrize("cfg", ["deblur_train", "sr2_train", "sr2_eval"])
def test_token_streaming_path_matches_the_tiled_path(cfg):
    """The bf16 SwinIR step on the token-streaming kernels (one launch for a block's four weight gradients, the layer's
    matrix in registers, LayerNorm forward / backward inside the GEMM epilogues, GELU' input recomputed) against the
    same step on the tiled GEMMs + separate LayerNorm / cast kernels (`_ops.TOKEN_STREAMING = False`), same weights,
    inputs and stochastic-depth masks: the two are different summation orders of the same bf16 products."""
    from models import _ops
    from models.swinir import SwinIR
    up = 2 if cfg.startswith("sr2") else 1
    depths = (3, 2)
    torch.manual_seed(11)
    model = SwinIR(upscale=up, upsampler="pixelshuffle" if up > 1 else None, depths=depths, num_heads=(6, 6)).cuda()
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.endswith("bias") or "norm" in k:
                v.add_(0.1 * torch.randn(v.shape, generator=gen).cuda())
    model.train(cfg.endswith("train"))
    x = torch.rand((2, 3, 32, 32), generator=gen).cuda()          # 2048 tokens: a multiple of 64 (the streaming kernels run)
    go = torch.randn((2, 3, 32 * up, 32 * up), generator=gen).cuda()
    masks = None
    if model.training:
        masks = [None if m is None else tuple(v.cuda() for v in m)
                 for m in sp.draw_drop_masks(2, depths=depths, rate=0.4, generator=gen)]
    outs = {}
    prev = _ops.set_compute_dtype("bf16")
    try:
        for streaming in (True, False):
            _ops.TOKEN_STREAMING = streaming
            with N_call_log() as log:
                model.zero_grad_flat()
                out = model(x, drop_masks=masks)
                out.backward(go)
                torch.cuda.synchronize()
            names = {n for n, _ in log}
            new = {"sei_rowgemm_bf16", "sei_rowgemm_ln_bf16", "sei_rowgemm_lnbwd_bf16", "sei_rowgemm_dgelu_bf16",
                   "sei_tokgrad_bf16_blocks"}
            assert (new <= names) if streaming else not (new & names), sorted(names)
            outs[streaming] = (out.detach().clone(), model.flat_grads.clone())
    finally:
        _ops.TOKEN_STREAMING = True
        _ops.set_compute_dtype(prev)
    assert relerr(outs[True][0], outs[False][0]) < 5e-3, relerr(outs[True][0], outs[False][0])
    base = model.flat_params.data_ptr()
    for k, p in model.named_parameters():
        off = (p.data_ptr() - base) // 4
        a, b = outs[True][1][off:off + p.numel()].double(), outs[False][1][off:off + p.numel()].double()
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, k
            continue
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        # Every bf16 materialisation point (LayerNorm outputs, qkv, attention output, GELU output, gradients) rounds on a
        # different side in the two paths: 0.9988-0.9995 per parameter here (either path against exact float32: ~0.98-0.999,
        # test_swinir_bf16_path_tracks_f32); a wrong index map or a dropped term shows as a cosine far below
        assert cos > 0.997 and abs(float(a.norm() / b.norm()) - 1) < 3e-2, (k, cos, float(a.norm()), float(b.norm()))


class N_call_log:
    """Record the entry points called through _native.call inside the block."""

    def __enter__(self):
        import _native
        self._n = _native
        self._prev = _native._CALL_LOG
        _native._CALL_LOG = []
        return _native._CALL_LOG

    def __exit__(self, *exc):
        self._n._CALL_LOG = self._prev
        return False


def test_conv_bias_gradient_in_a_ones_column_of_the_input_grid():
    """sei_pad_nhwc_bf16_ones: channel C of the padded grid is 1.0 in every row (zeros elsewhere in the padding, the
    pixels as sei_pad_nhwc_bf16 writes them), and the tap-batched weight gradient of a convolution on that grid then
    holds the column sums of the output gradient -- the bias gradient -- in column C of every tap."""
    import ctypes
    import _native as N
    B, H, W, Cin, Cout, cinp, coutp = 2, 16, 24, 180, 180, 192, 192
    Wp, R = W + 2, B * (H + 2) * (W + 2)
    guard, R8 = Wp + 9, (R + 7) // 8 * 8
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, H, W, Cin), device="cuda", generator=gen)
    gy = torch.randn((B, H, W, Cout), device="cuda", generator=gen)
    xp0 = torch.empty((R + 2 * guard, cinp), device="cuda", dtype=torch.bfloat16)
    xp1 = torch.empty_like(xp0)
    gop = torch.empty((R + 2 * guard, coutp), device="cuda", dtype=torch.bfloat16)
    N.call("sei_pad_nhwc_bf16", x.data_ptr(), xp0.data_ptr(), B, H, W, Cin, cinp, guard)
    N.call("sei_pad_nhwc_bf16_ones", x.data_ptr(), xp1.data_ptr(), B, H, W, Cin, cinp, guard, 1)
    N.call("sei_pad_nhwc_bf16", gy.data_ptr(), gop.data_ptr(), B, H, W, Cout, coutp, guard)
    assert torch.equal(xp1[:, :Cin], xp0[:, :Cin]) and torch.equal(xp1[:, Cin + 1:], xp0[:, Cin + 1:])
    assert bool((xp1[:, Cin].float() == 1.0).all()) and bool((xp0[:, Cin].float() == 0.0).all())
    with pytest.raises(N.NativeLibraryError):             # no padding channel to put the ones in
        N.call("sei_pad_nhwc_bf16_ones", x.data_ptr(), xp1.data_ptr(), B, H, W, Cin, Cin, guard, 1)
    offs = (ctypes.c_int * 9)(*[(ky - 1) * Wp + (kx - 1) for ky in range(3) for kx in range(3)])
    taps = torch.zeros((9, coutp, cinp), device="cuda")
    g, xg = gop[guard:guard + R8], xp1[guard:guard + R8]
    N.call("sei_gemm_bf16nt_dw2_taps", g.data_ptr(), g.data_ptr(), coutp, xg.data_ptr(), xg.data_ptr(), cinp, taps.data_ptr(),
           coutp, cinp, R8, 0, 1, 9, ctypes.cast(offs, ctypes.c_void_p), coutp * cinp)
    want = gy.bfloat16().double().sum((0, 1, 2))          # what the GEMM sums: the bf16-rounded gradient
    for t in range(9):
        assert relerr(taps[t, :Cout, Cin], want) < 1e-5
    assert relerr(taps[4, :Cout, Cin], gy.double().sum((0, 1, 2))) < 3e-3      # against the float32 column sums it replaces
