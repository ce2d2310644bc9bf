"""Paired cropping (reference call surface: src/crop.py), index arithmetic only.

`MinSizePadding` reads `x.shape[1]` and `x.shape[2]` as (height, width). That is right for the 3-D
dataset items it was written for, but `Loss.forward` (src/losses/__init__.py:203-205) also feeds it
4-D batches, where those are (C, H): the batch then gets `size - C` zero rows appended and the crop
offset is drawn over the padded height. The default here reproduces that, because it
changes which pixels are trained on; `fix_batched_crop=True` (flag --fix_batched_crop of train.py)
reads the last two dimensions instead.
"""
from math import ceil

import torch
import torch.nn.functional as F
from torch.nn import Module

FIX_BATCHED_CROP = False     # process-wide default; train.py sets it from --fix_batched_crop


class MinSizePadding(Module):
    def __init__(self, size, padding_mode="constant", fill=0, fix_batched_crop=None):
        super().__init__()
        if padding_mode != "constant":
            raise ValueError("only constant padding is supported")
        self.size = size
        self.padding_mode = padding_mode
        self.fill = fill
        self.fix_batched_crop = fix_batched_crop

    def forward(self, x):
        fix = FIX_BATCHED_CROP if self.fix_batched_crop is None else self.fix_batched_crop
        h, w = (x.shape[-2], x.shape[-1]) if fix else (x.shape[1], x.shape[2])
        pad_bottom, pad_right = max(0, self.size - h), max(0, self.size - w)
        if pad_bottom == 0 and pad_right == 0:
            return x
        return F.pad(x, (0, pad_right, 0, pad_bottom), value=self.fill)


class CropPair(Module):
    def __init__(self, location, size):
        super().__init__()
        assert location in ["random", "center"]
        self.location = location
        self.size = size

    def draw_offsets(self, y_shape):
        """(i, j, h, w): the crop offset `forward` would draw for a y of this shape -- the same two draws from the CPU
        generator, over the MinSizePadding-padded extent (h, w) -- without touching any data. For callers that need the
        cropped y only (graphs.GraphedLossStep with a loss that never reads x): `write_y` then fills the crop."""
        s = self.size
        fix = FIX_BATCHED_CROP
        h0, w0 = (y_shape[-2], y_shape[-1]) if fix else (y_shape[1], y_shape[2])
        h, w = y_shape[-2] + max(0, s - h0), y_shape[-1] + max(0, s - w0)
        if self.location == "random":
            i = torch.randint(0, h - s + 1, size=(1,)).item()
            j = torch.randint(0, w - s + 1, size=(1,)).item()
        else:
            i, j = (h - s) // 2, (w - s) // 2
        return i, j, h, w

    def write_y(self, y, i, j, out):
        """out <- the (size x size) crop of the zero-padded y at (i, j): one strided copy when the window lies inside y,
        a zero fill in front of it when it reaches into the padding (the batched-crop quirk, module docstring)."""
        s = self.size
        if y.is_cuda and y.dtype == torch.float32 and out.dtype == torch.float32 and y.is_contiguous() \
                and out.is_contiguous() and y.dim() >= 2 and tuple(out.shape) == tuple(y.shape[:-2]) + (s, s):
            import _native as N                         # one launch: the window, zeros where it leaves y
            N.call("sei_crop_window", y.data_ptr(), out.data_ptr(), y.numel() // (y.shape[-2] * y.shape[-1]), y.shape[-2],
                   y.shape[-1], int(i), int(j), s)
            return out
        vh, vw = min(s, y.shape[-2] - i), min(s, y.shape[-1] - j)
        if vh < s or vw < s:
            out.zero_()
        if vh > 0 and vw > 0:
            out[..., :vh, :vw].copy_(y[..., i:i + vh, j:j + vw])
        return out

    def forward(self, x, y, xy_size_ratio=None):
        if xy_size_ratio is None:
            xy_size_ratio = int(ceil(x.shape[1] / y.shape[1]))
        r, s = xy_size_ratio, self.size
        x = MinSizePadding(s * r)(x)
        y = MinSizePadding(s)(y)
        h, w = y.shape[-2:]
        if self.location == "random":
            # two draws from the CPU generator, in this order, as the reference (:26-27)
            i = torch.randint(0, h - s + 1, size=(1,)).item()
            j = torch.randint(0, w - s + 1, size=(1,)).item()
        else:
            i, j = (h - s) // 2, (w - s) // 2
        return (x[..., i * r:(i + s) * r, j * r:(j + s) * r], y[..., i:i + s, j:j + s])
