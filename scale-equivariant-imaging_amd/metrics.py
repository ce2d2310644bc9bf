"""Evaluation metric of the path: PSNR on the luma channel (reference: src/metrics.py).

`psnr_fn` restates kornia.color.rgb_to_ycbcr's Y (0.299 R + 0.587 G + 0.114 B) and
torchmetrics.functional.peak_signal_noise_ratio(data_range=1.0) = 10 log10(1 / mse); both libraries are absent
here, so the restatement is pinned only against the oracle's own (oracle/torch_path.py: psnr_y) -- SURVEY 8c
"parity unpinned" for this row. On GPU tensors the squared-error sum runs in sei_luma_sqerr. SSIM and LPIPS
(torchmetrics / pyiqa) are not rebuilt: `compute_metrics` returns NaN for them.
"""
import math

import torch


def luma(img):
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    return 0.299 * r + 0.587 * g + 0.114 * b


def psnr_fn(x_hat, x):
    """x_hat, x: (3, H, W) in [0, 1] -> scalar tensor (dB)."""
    if x_hat.is_cuda and x.is_cuda and x_hat.dtype == torch.float32 and x.dtype == torch.float32:
        import _native as N
        a, b = x_hat.contiguous(), x.contiguous()
        npix = a.shape[-2] * a.shape[-1]
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        work = torch.empty(256, dtype=torch.float32, device=a.device)
        N.call("sei_luma_sqerr", a.data_ptr(), b.data_ptr(), npix, out.data_ptr(), work.data_ptr())
        err = out[0] / npix
    else:
        err = (luma(x_hat) - luma(x)).pow(2).mean()
    return 10.0 * torch.log10(1.0 / err)


def register_fn(x, x_hat):
    """Centre-crop both images to their common size (reference :33-40, torchvision CenterCrop)."""
    if x.shape[-2] != x_hat.shape[-2] or x.shape[-1] != x_hat.shape[-1]:
        hmin, wmin = min(x.shape[-2], x_hat.shape[-2]), min(x.shape[-1], x_hat.shape[-1])

        def centre(t):
            top = int(round((t.shape[-2] - hmin) / 2.0))
            left = int(round((t.shape[-1] - wmin) / 2.0))
            return t[..., top:top + hmin, left:left + wmin]
        x, x_hat = centre(x), centre(x_hat)
    return x, x_hat


def compute_metrics(x, x_hat):
    """(psnr, ssim, lpips) as the reference's compute_metrics; only the PSNR is computed here."""
    x, x_hat = register_fn(x, x_hat)
    return psnr_fn(x, x_hat).item(), math.nan, math.nan
