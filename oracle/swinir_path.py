"""ORACLE (test infrastructure) -- torch-CPU restatement of SwinIR as the reference configures it.

PARITY UNPINNED: the reference's default backbone is `deepinv.models.SwinIR` (deepinv v0.2.0, which vendors the
official SwinIR `network_swinir.py` and imports `DropPath` / `trunc_normal_` from timm 0.9.12). Neither package is in
the reference tree or importable here, and the reference holds no fixtures for it, so this file restates the
PUBLISHED architecture (Liang et al., "SwinIR: Image Restoration Using Swin Transformer", ICCVW 2021, and the
official implementation's module tree / state_dict keys) for the constructor arguments at
/root/reference/src/models/__init__.py:51-74:

    upscale = 1 (deblurring) | sr_factor, upsampler = None | "pixelshuffle", img_size 48, patch_size 1, in_chans 3,
    embed_dim 180, depths [6]*6, num_heads [6]*6, window_size 8, mlp_ratio 2, qkv_bias True, qk_scale None,
    drop_rate 0, attn_drop_rate 0, drop_path_rate 0.1, LayerNorm, ape False, patch_norm True, img_range 1.0,
    resi_connection "1conv"

Functional over a flat state_dict with the official key layout (`conv_first`, `patch_embed.norm`,
`layers.{i}.residual_group.blocks.{j}.{norm1,attn.{relative_position_bias_table,qkv,proj},norm2,mlp.{fc1,fc2}}`,
`layers.{i}.conv`, `norm`, `conv_after_body`, `conv_before_upsample.0`, `upsample.{0,2}`, `conv_last`); the
`conv_last.*` keys are what demo/train.py:180-184 fine-tunes. Stochastic depth (timm DropPath: one Bernoulli(keep)
per sample and per call, divided by keep) is injectable through `drop_masks`.
"""
import math

import torch
import torch.nn.functional as F

EMBED, HEADS, WINDOW, DEPTHS, MLP_RATIO, DROP_PATH, NUM_FEAT = 180, 6, 8, (6, 6, 6, 6, 6, 6), 2, 0.1, 64
RGB_MEAN = (0.4488, 0.4371, 0.4040)


def drop_path_rates(depths=DEPTHS, rate=DROP_PATH):
    """network_swinir.py: dpr = linspace(0, drop_path_rate, sum(depths)), one rate per block."""
    return [float(v) for v in torch.linspace(0, rate, sum(depths))]


def relative_position_index(ws=WINDOW):
    """WindowAttention.__init__: (ws*ws, ws*ws) indices into the (2ws-1)^2-row bias table."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(H, W, ws=WINDOW, shift=WINDOW // 2):
    """SwinTransformerBlock.calculate_mask: (nW, ws*ws, ws*ws) with 0 / -100."""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _window_attention(sd, p, xw, mask, heads=HEADS):
    B_, N, C = xw.shape
    qkv = F.linear(xw, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(int(math.isqrt(N)))
    bias = sd[p + "relative_position_bias_table"][idx.view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def _block(sd, p, x, x_size, shift, drop=None, ws=WINDOW):
    """SwinTransformerBlock.forward; drop = (mask_attn, mask_mlp) per-sample factors (B,) or None."""
    H, W = x_size
    B, L, C = x.shape
    shortcut = x
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]).view(B, H, W, C)
    if shift:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(h, ws).view(-1, ws * ws, C)
    aw = _window_attention(sd, p + "attn.", xw, shift_mask(H, W, ws, shift) if shift else None)
    h = window_reverse(aw.view(-1, ws, ws, C), ws, H, W)
    if shift:
        h = torch.roll(h, shifts=(shift, shift), dims=(1, 2))
    h = h.view(B, H * W, C)
    if drop is not None:
        h = h * drop[0].to(h.dtype).view(B, 1, 1)
    x = shortcut + h
    m = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    m = F.linear(F.gelu(F.linear(m, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                 sd[p + "mlp.fc2.bias"])
    if drop is not None:
        m = m * drop[1].to(m.dtype).view(B, 1, 1)
    return x + m


def draw_drop_masks(batch, training=True, depths=DEPTHS, rate=DROP_PATH, generator=None, dtype=torch.float32):
    """timm DropPath in call order (block by block: attention branch, then MLP branch): per-sample
    Bernoulli(keep) / keep; blocks whose rate is 0 are nn.Identity (no draw). None in eval mode."""
    if not training:
        return None
    masks = []
    for r in drop_path_rates(depths, rate):
        if r == 0.0:
            masks.append(None)
            continue
        keep = 1.0 - r
        pair = tuple(torch.empty(batch, dtype=dtype).bernoulli_(keep, generator=generator) / keep for _ in range(2))
        masks.append(pair)
    return masks


def swinir_forward(sd, x, *, upscale=1, drop_masks=None, depths=DEPTHS, ws=WINDOW, img_range=1.0):
    """SwinIR.forward (network_swinir.py): mean shift, reflect-pad to a multiple of the window, conv_first,
    RSTBs (+ 3x3 conv and residual each), norm, conv_after_body + residual, reconstruction head, un-shift, crop."""
    H, W = x.shape[2:]
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    x = F.pad(x, (0, pw, 0, ph), "reflect")
    mean = torch.tensor(RGB_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    x = (x - mean) * img_range
    first = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    Hp, Wp = first.shape[2:]
    C = first.shape[1]
    t = first.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"])
    blk = 0
    for i, depth in enumerate(depths):
        res = t
        for j in range(depth):
            drop = drop_masks[blk] if drop_masks is not None else None
            t = _block(sd, f"layers.{i}.residual_group.blocks.{j}.", t, (Hp, Wp), 0 if j % 2 == 0 else ws // 2, drop, ws)
            blk += 1
        img = t.transpose(1, 2).reshape(-1, C, Hp, Wp)
        img = F.conv2d(img, sd[f"layers.{i}.conv.weight"], sd[f"layers.{i}.conv.bias"], padding=1)
        t = img.flatten(2).transpose(1, 2) + res
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"])
    body = t.transpose(1, 2).reshape(-1, C, Hp, Wp)
    feat = F.conv2d(body, sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1) + first
    if upscale == 1:                                    # upsampler None: "for image denoising" branch
        out = x + F.conv2d(feat, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    else:                                               # "pixelshuffle"
        f = F.leaky_relu(F.conv2d(feat, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"],
                                  padding=1), 0.01)
        if upscale & (upscale - 1) == 0:
            for s in range(int(math.log2(upscale))):
                f = F.pixel_shuffle(F.conv2d(f, sd[f"upsample.{2 * s}.weight"], sd[f"upsample.{2 * s}.bias"], padding=1), 2)
        elif upscale == 3:
            f = F.pixel_shuffle(F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"], padding=1), 3)
        else:
            raise ValueError(f"scale {upscale} is not supported. Supported scales: 2^n and 3.")
        out = F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    out = out / img_range + mean
    return out[:, :, :H * upscale, :W * upscale]


def swinir_init_state_dict(upscale=1, seed=0, dtype=torch.float32, depths=DEPTHS, embed=EMBED, heads=HEADS, ws=WINDOW):
    """A state_dict with the official key layout and init distributions (Linear / bias table: trunc_normal(std .02),
    LayerNorm ones / zeros, Conv2d: torch default), seeded. Draw ORDER differs from nn.Module construction: for
    tests and benchmarks that share the dict with the product, not a reproduction of a seeded module init."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k=3):
        bound = 1.0 / (cin * k * k) ** 0.5
        sd[name + ".weight"] = ((torch.rand((cout, cin, k, k), generator=g) * 2 - 1) * bound).to(dtype)
        sd[name + ".bias"] = ((torch.rand((cout,), generator=g) * 2 - 1) * bound).to(dtype)

    def linear(name, cout, cin):
        w = torch.empty(cout, cin)
        torch.nn.init.trunc_normal_(w, std=0.02, generator=g)
        sd[name + ".weight"] = w.to(dtype)
        sd[name + ".bias"] = torch.zeros(cout, dtype=dtype)

    def ln(name):
        sd[name + ".weight"] = torch.ones(embed, dtype=dtype)
        sd[name + ".bias"] = torch.zeros(embed, dtype=dtype)

    conv("conv_first", embed, 3)
    ln("patch_embed.norm")
    hidden = int(embed * MLP_RATIO)
    for i, depth in enumerate(depths):
        for j in range(depth):
            p = f"layers.{i}.residual_group.blocks.{j}."
            ln(p + "norm1")
            t = torch.empty((2 * ws - 1) ** 2, heads)
            torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
            sd[p + "attn.relative_position_bias_table"] = t.to(dtype)
            linear(p + "attn.qkv", 3 * embed, embed)
            linear(p + "attn.proj", embed, embed)
            ln(p + "norm2")
            linear(p + "mlp.fc1", hidden, embed)
            linear(p + "mlp.fc2", embed, hidden)
        conv(f"layers.{i}.conv", embed, embed)
    ln("norm")
    conv("conv_after_body", embed, embed)
    if upscale == 1:
        conv("conv_last", 3, embed)
    else:
        conv("conv_before_upsample.0", NUM_FEAT, embed)
        if upscale & (upscale - 1) == 0:
            for s in range(int(math.log2(upscale))):
                conv(f"upsample.{2 * s}", 4 * NUM_FEAT, NUM_FEAT)
        elif upscale == 3:
            conv("upsample.0", 9 * NUM_FEAT, NUM_FEAT)
        else:
            raise ValueError(f"scale {upscale} is not supported. Supported scales: 2^n and 3.")
        conv("conv_last", 3, NUM_FEAT)
    return sd
