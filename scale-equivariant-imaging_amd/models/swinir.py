"""SwinIR backbone (the reference's default `--ProposedModel__architecture Transformer`:
`deepinv.models.SwinIR(...)` with the arguments at src/models/__init__.py:51-74).

deepinv v0.2.0 vendors the official SwinIR `network_swinir.py` (Liang et al., ICCVW 2021) and takes DropPath /
trunc_normal_ from timm; neither package is part of the reference tree or installed here, so this module is built
from the PUBLISHED architecture: same module tree, parameter / buffer names and shapes (so that `state_dict()`
interchanges with the published weights and `model.model.conv_last.*`, which demo/train.py:180-184 fine-tunes, is
where the reference expects it), same construction order and init calls (so a seeded construction draws the same
numbers from the same torch calls). PARITY UNPINNED beyond that: the checker is oracle/swinir_path.py, a restatement
of the same published architecture, not an output of the reference.

torch layer classes are PARAMETER CONTAINERS only; every forward / backward runs in libsei_hip.so
(models/_swin_ops.py) on tokens stored as one (B*H*W, C) matrix in natural order = the NHWC image.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

import _native as N
from physics._ops import axpy
from . import _ops, _swin_ops as S, _swin_ops16 as S16
from ._flat import FlatParameterBucket

RGB_MEAN = (0.4488, 0.4371, 0.4040)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))       # in the state_dict, as upstream
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class DropMasks(list):
    """The stochastic-depth factors of one model call: a list with one (attention branch, MLP branch) pair of per-sample
    vectors per block (None for a block that draws nothing), all of them rows of `buffer` (2 x drawing blocks, batch)."""

    def __init__(self, buffer):
        super().__init__()
        self.buffer = buffer

    def rebuilt_on(self, buffer):
        """The same structure over another buffer of the same row count (concatenated batches, a static copy)."""
        out, j = DropMasks(buffer), 0
        for m in self:
            if m is None:
                out.append(None)
            else:
                out.append((buffer[j], buffer[j + 1]))
                j += 2
        return out

    def rows(self, tokens_per_image):
        """Per-ROW factors (what the GEMM epilogues take), all blocks expanded by one launch."""
        wide = self.buffer.to(torch.float32).repeat_interleave(tokens_per_image, dim=1)
        return [None if m is None else (m[0], m[1]) for m in self.rebuilt_on(wide)]


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio, drop_path):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.drop_path_rate = window_size, shift_size, float(drop_path)
        if min(input_resolution) <= window_size:
            raise ValueError("input_resolution must exceed the window (the reference's img_size is 48, window 8)")
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        mask = self.calculate_mask(input_resolution) if shift_size > 0 else None
        self.register_buffer("attn_mask", mask)                             # state_dict entry of shifted blocks

    def calculate_mask(self, x_size):
        """The reference's mask tensor (kept for the state_dict; the kernel derives it from coordinates)."""
        H, W = x_size
        ws, sh = self.window_size, self.shift_size
        img = torch.zeros((1, H, W, 1))
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
        m = mw.unsqueeze(1) - mw.unsqueeze(2)
        return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)

    def forward(self, x, drop=None, pack=None, pre=None, nxt=None, prev_scale=None):
        """x: (B, H, W, C) tokens; drop: (rows_attn, rows_mlp) per-row stochastic-depth factors or None. bf16 mode only:
        pre = this block's norm1 output / statistics already formed by the previous block, nxt = the next block's norm1
        (weight, bias) to apply to this block's output -- the return value is then (out, h, mean, rstd)."""
        a, m = self.attn, self.mlp
        d1, d2 = drop if drop is not None else (None, None)
        if pack is not None:                            # throughput mode: bf16 GEMM layouts from the pack
            return S16.SwinBlockFn16.apply(x, self.norm1.weight, self.norm1.bias, a.relative_position_bias_table,
                                           a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.bias, m.fc2.bias,
                                           pack, self._sei_key, self.num_heads, self.shift_size, d1, d2, pre, nxt,
                                           prev_scale)
        return S.SwinBlockFn.apply(x, self.norm1.weight, self.norm1.bias, a.relative_position_bias_table,
                                   a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, self.norm2.weight,
                                   self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                   self.num_heads, self.shift_size, d1, d2)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, drop_path):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, drop_path[i])
            for i in range(depth)])


class PatchEmbed(nn.Module):
    """Parameter container of the reference's PatchEmbed (only its optional LayerNorm has parameters)."""

    def __init__(self, embed_dim, norm):
        super().__init__()
        self.norm = nn.LayerNorm(embed_dim) if norm else None


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, drop_path):
        super().__init__()
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio, drop_path)
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)

    def forward(self, x, drops, pack=None):
        res = x
        blocks = list(self.residual_group.blocks)
        pre = prev_scale = None
        for i, (blk, drop) in enumerate(zip(blocks, drops)):
            if pack is None:
                x = blk(x, drop, pack)
                continue
            # bf16 mode: a block's last launch also applies the NEXT block's norm1 to the rows it has just formed, and
            # its norm1 backward leaves the bf16 operand the PREVIOUS block's backward starts from (prev_scale)
            if i + 1 < len(blocks):
                nxt = (blocks[i + 1].norm1.weight, blocks[i + 1].norm1.bias)
                x, *pre = blk(x, drop, pack, pre, nxt, prev_scale)
            else:
                x, pre = blk(x, drop, pack, pre, None, prev_scale), None
            prev_scale = drop[1] if drop is not None else None
        return conv3x3(self.conv, x, res, 0, pack)


def conv3x3(conv, x, res, act, pack):
    """A many-channel 3x3 convolution of the backbone on NHWC tokens: the f32 padded-grid form, or (pack given) the
    bf16 implicit GEMM."""
    if pack is not None:
        return S16.Conv3x3GemmFn16.apply(x, conv.weight, conv.bias, res, act, pack, conv._sei_key)
    return S.Conv3x3GemmFn.apply(x, conv.weight, conv.bias, res, act)


class SwinIR(FlatParameterBucket, nn.Module):
    def __init__(self, img_size=48, patch_size=1, in_chans=3, embed_dim=180, depths=(6, 6, 6, 6, 6, 6),
                 num_heads=(6, 6, 6, 6, 6, 6), window_size=8, mlp_ratio=2, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, upscale=1, img_range=1.0, upsampler=None, resi_connection="1conv",
                 pretrained=None):
        super().__init__()
        if (patch_size != 1 or not qkv_bias or qk_scale is not None or drop_rate or attn_drop_rate or ape
                or norm_layer is not nn.LayerNorm or resi_connection != "1conv" or window_size != 8
                or pretrained is not None or in_chans != 3 or upsampler not in (None, "", "pixelshuffle")
                or img_range != 1.0 or use_checkpoint):
            raise NotImplementedError("this build implements SwinIR for the reference's configuration "
                                      "(src/models/__init__.py:51-74); see models/swinir.py")
        if embed_dim % 4 or any(embed_dim % h or (embed_dim // h) not in (8, 16, 30, 32) for h in num_heads):
            raise NotImplementedError("head_dim must be 8, 16, 30 or 32 and embed_dim a multiple of 4")
        num_feat = 64
        self.img_range, self.upscale, self.upsampler, self.window_size = img_range, upscale, upsampler, window_size
        self.embed_dim = embed_dim
        self.register_buffer("mean", torch.tensor(RGB_MEAN).view(1, 3, 1, 1), persistent=False)
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(embed_dim, patch_norm)
        resolution = (img_size, img_size)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList([
            RSTB(embed_dim, resolution, depths[i], num_heads[i], window_size, mlp_ratio,
                 dpr[sum(depths[:i]):sum(depths[:i + 1])]) for i in range(len(depths))])
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        if upsampler == "pixelshuffle":
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            stages = []
            if upscale & (upscale - 1) == 0:
                for _ in range(int(math.log2(upscale))):
                    stages += [nn.Conv2d(num_feat, 4 * num_feat, 3, 1, 1), nn.PixelShuffle(2)]
            elif upscale == 3:
                stages += [nn.Conv2d(num_feat, 9 * num_feat, 3, 1, 1), nn.PixelShuffle(3)]
            else:
                raise ValueError(f"scale {upscale} is not supported. Supported scales: 2^n and 3.")
            self.upsample = nn.Sequential(*stages)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
        else:
            self.conv_last = nn.Conv2d(embed_dim, in_chans, 3, 1, 1)
        self.apply(self._init_weights)
        for name, module in self.named_modules():
            module._sei_key = name                      # key of the module's matrices in the bf16 pack
        self._pack = None
        self._init_bucket()

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # -- stochastic depth ------------------------------------------------------------------------
    def blocks(self):
        return [blk for layer in self.layers for blk in layer.residual_group.blocks]

    def draw_drop_masks(self, batch, device=None):
        """timm DropPath in call order (block by block: attention branch, then MLP branch): per-sample
        Bernoulli(keep) / keep from the global generator of `device`; None for a block whose rate is 0 (nn.Identity
        upstream: no draw) and in eval mode."""
        if not self.training:
            return None
        device = device if device is not None else self.conv_first.weight.device
        blocks = list(self.blocks())
        # the same draws as before (one bernoulli_ + div_ per DropPath call, in call order), into the rows of ONE buffer:
        # a captured step then refreshes its masks with one copy and expands them to rows with one launch
        keeps = [1.0 - b.drop_path_rate for b in blocks if b.drop_path_rate != 0.0 for _ in range(2)]
        masks = DropMasks(torch.empty((len(keeps), batch), device=device))
        j = 0
        for blk in blocks:
            if blk.drop_path_rate == 0.0:
                masks.append(None)
            else:
                masks.append(tuple(masks.buffer[j + i].bernoulli_(keeps[j + i]) for i in range(2)))
                j += 2
        if keeps:                                        # x / keep for every row at once (the same IEEE division as div_)
            cache = self.__dict__.setdefault("_keep_columns", {})
            col = cache.get((device, len(keeps)))
            if col is None:
                col = cache[(device, len(keeps))] = torch.tensor(keeps, dtype=torch.float32, device=device)[:, None]
            masks.buffer.div_(col)
        return masks

    def _last(self, t, res, pack):
        """conv_last on NHWC tokens -> NCHW image (+ res): the small direct kernel, or in throughput mode the implicit
        GEMM with a zero 4th output channel that is sliced away."""
        if pack is None:
            return _ops.Conv3x3Fn.apply(t, self.conv_last.weight, self.conv_last.bias, res, False, True)
        y = conv3x3(self.conv_last, t, None, 0, pack)[..., :3].permute(0, 3, 1, 2).contiguous()
        return y if res is None else axpy(y, res, 1.0)

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, x, drop_masks="draw"):
        """x: (B, 3, H, W). drop_masks: "draw" (training: draw them here, eval: none), None, or the list returned by
        draw_drop_masks (injected by tests / a captured step)."""
        with _ops.compute_dtype_scope(self):             # this model's own arithmetic mode, if it carries one
            return self._forward(x, drop_masks)

    def _forward(self, x, drop_masks):
        x = N.check_tensor(x.contiguous(), "x")
        _ops.note_forward(self)
        B, _, H, W = x.shape
        ws = self.window_size
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph), "reflect")
        Hp, Wp = H + ph, W + pw
        if isinstance(drop_masks, str):
            drop_masks = self.draw_drop_masks(B, x.device)
        pack = None
        if _ops.get_compute_dtype() == "bf16":           # throughput mode: re-lay out the weights for the bf16 GEMMs
            if self._pack is None or not self._pack.valid_for(self):
                self._pack = S16.SwinPack(self)
            pack = self._pack
            pack.refresh()
        mean = self.mean.to(x.dtype).expand(B, 3, Hp, Wp).contiguous()
        x = axpy(x.contiguous(), mean, -1.0)             # (x - mean) * img_range, img_range = 1
        if pack is not None:                            # NHWC image with a zero 4th channel (data movement only)
            first = conv3x3(self.conv_first, F.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous(), None, 0, pack)
        else:
            first = _ops.Conv3x3Fn.apply(x, self.conv_first.weight, self.conv_first.bias, None, True, False)   # -> NHWC
        t = first
        if self.patch_embed.norm is not None:
            t = S.LayerNormFn.apply(t, self.patch_embed.norm.weight, self.patch_embed.norm.bias)
        k = 0
        row_masks = drop_masks.rows(Hp * Wp) if isinstance(drop_masks, DropMasks) else None
        for layer in self.layers:
            n = len(layer.residual_group.blocks)
            drops = [None] * n
            if row_masks is not None:
                drops = row_masks[k:k + n]
            elif drop_masks is not None:
                drops = [None if m is None else tuple(v.to(torch.float32).repeat_interleave(Hp * Wp) for v in m)
                         for m in drop_masks[k:k + n]]
            t = layer(t, drops, pack)
            k += n
        t = S.LayerNormFn.apply(t, self.norm.weight, self.norm.bias)
        feat = conv3x3(self.conv_after_body, t, first, 0, pack)
        if self.upsampler == "pixelshuffle":
            c0 = self.conv_before_upsample[0]
            f = conv3x3(c0, feat, None, 1, pack)
            for stage in self.upsample:
                if isinstance(stage, nn.Conv2d):
                    f = conv3x3(stage, f.contiguous(), None, 0, pack)
                else:                                   # PixelShuffle on NHWC: a pure permutation (data movement)
                    r = stage.upscale_factor
                    Bf, Hf, Wf, Cf = f.shape
                    f = f.view(Bf, Hf, Wf, Cf // (r * r), r, r).permute(0, 1, 4, 2, 5, 3).reshape(Bf, Hf * r, Wf * r,
                                                                                                Cf // (r * r))
            out = self._last(f.contiguous(), None, pack)
            mean_out = self.mean.to(x.dtype).expand(B, 3, Hp * self.upscale, Wp * self.upscale).contiguous()
        else:
            out = self._last(feat, x, pack)
            mean_out = mean
        out = axpy(out, mean_out, 1.0)                   # x / img_range + mean
        return out[:, :, :H * self.upscale, :W * self.upscale].contiguous()
