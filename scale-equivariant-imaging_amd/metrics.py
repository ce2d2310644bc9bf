"""PSNR on the luma channel (reference: src/metrics.py:10-13 -- kornia rgb_to_ycbcr + torchmetrics
peak_signal_noise_ratio with data_range=1; both restated). SSIM/LPIPS are outside scope."""
import torch


def luma(img):
    r, g, b = img[..., 0, :, :], img[..., 1, :, :], img[..., 2, :, :]
    return 0.299 * r + 0.587 * g + 0.114 * b


def psnr_fn(x_hat, x):
    """x_hat, x: (3, H, W) in [0, 1] -> scalar tensor (dB)."""
    err = (luma(x_hat) - luma(x)).pow(2).mean()
    return 10.0 * torch.log10(1.0 / err)
