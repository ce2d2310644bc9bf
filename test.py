#!/usr/bin/env python3
"""Evaluation driver with the reference's command line (reference: demo/test.py).

    python test.py --device cuda --task deblurring --kernel Gaussian_R2 --ProposedModel__architecture Convolutional \
        --dataset div2k --GroundTruthDataset__datasets_dir ./datasets --weights runs/x/weights.pt

For every test pair: x_hat = model(y) under no_grad (the same HIP forward as training, any image size), then
quantise to 8 bits and clamp x, y, x_hat (reference :140-148), PSNR on the luma channel (src/metrics.py), and
the reference's summary lines. In scope: the Proposed model family, `--dataset div2k | single_image | synthetic`
or a directory of PNG measurements, `--save_images`, `--save_psf`, `--indices`, `--print_all_metrics`,
`--noise2inverse` (src/noise2inverse.py's sliced evaluation around the same backbone) and `--r2r`.
Out of scope and refused: DIP / PnP / BM3D / DiffPIR / DPS / TV baselines; SSIM and LPIPS are
printed as nan (torchmetrics / pyiqa are not rebuilt).
"""
import os
import sys
from argparse import BooleanOptionalAction
from os.path import basename, dirname, isdir

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scale-equivariant-imaging_amd"))

from datasets import get_dataset  # noqa: E402
from datasets._io import read_image  # noqa: E402
from metrics import compute_metrics  # noqa: E402
from models import get_model  # noqa: E402
from physics import get_physics  # noqa: E402
from settings import DefaultArgParser  # noqa: E402
from training import get_weights  # noqa: E402


def build_parser():
    parser = DefaultArgParser()
    flag = parser.add_argument
    flag("--weights", type=str)
    flag("--save_images", action="store_true")
    flag("--indices", type=str, default=None)
    flag("--out_dir", type=str, default=None)
    flag("--save_psf", action="store_true")
    flag("--dip_iterations", type=int, default=None)
    flag("--noise2inverse", action="store_true")
    flag("--print_all_metrics", action="store_true")
    flag("--r2r", action="store_true")
    flag("--r2r_itercount", type=int, default=1)
    flag("--tv_lambd", type=float, default=None)
    flag("--tv_max_iter", type=int, default=300)
    flag("--GroundTruthDataset__split", type=str, default="val")
    flag("--SyntheticDataset__deterministic_measurements", action=BooleanOptionalAction, default=True)
    flag("--memoize_gt", action=BooleanOptionalAction, default=False)
    flag("--compute_dtype", choices=["f32", "bf16", "bf16x3"], default="f32")          # build-side addition
    return parser


def quantize_and_clamp(im):
    """8-bit quantisation, then clamping (reference :141-145)."""
    return ((im * 255.0).round() / 255.0).clamp(0.0, 1.0)


def save_image(im, path):
    """(1, C, H, W) or (C, H, W) in [0, 1] -> PNG (torchvision.utils.save_image for one image: x255 + 0.5, uint8)."""
    from PIL import Image
    im = im.detach()
    if im.dim() == 4:
        im = im[0]
    if im.dim() == 2:
        im = im[None]
    a = im.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(path)


def main(argv=None):
    torch.manual_seed(0)
    np.random.seed(0)
    args = build_parser().parse_args(argv)
    if args.model_kind == "dip":
        raise NotImplementedError("DIP evaluation is outside the hot path of this build")
    from models import _ops as model_ops
    model_ops.set_compute_dtype(args.compute_dtype)

    physics = None if isdir(args.dataset) else get_physics(args, device=args.device)
    model = get_model(args=args, physics=physics, device=args.device)
    model.to(args.device)
    model.eval()
    if args.weights is not None:
        model.load_weights(get_weights(args.weights, args.device))

    basename_table = {}
    if isdir(args.dataset):                                   # a folder of measurements, no ground truth
        from glob import glob
        dataset = []
        for i, f in enumerate(glob(os.path.join(args.dataset, "*.png"))):
            y = read_image(f).to(args.device).float() / 255.0
            dataset.append((None, y[:3, :, :]))               # discard the alpha channel if it exists
            basename_table[i] = basename(f)
    else:
        dataset = get_dataset(args=args, purpose="test", physics=physics, device=args.device, _HOTFIX=False)

    if args.save_psf:
        assert args.out_dir is not None
        assert physics.task == "deblurring"
        kernel = physics.filter
        assert kernel.dim() == 4
        kernel = kernel.squeeze(0).squeeze(0)
        os.makedirs(args.out_dir, exist_ok=True)
        save_image((kernel / kernel.max()).float(), os.path.join(args.out_dir, "psf.png"))

    indices = range(len(dataset)) if args.indices is None else (int(i) for i in args.indices.split(","))
    psnr_list, ssim_list, lpips_list = [], [], []
    for i in indices:
        x, y = dataset[i]
        x = x.unsqueeze(0) if x is not None else None
        y = y.unsqueeze(0)
        with torch.no_grad():
            if args.noise2inverse:                            # reference demo/test.py:116-126
                from noise2inverse import Noise2InverseModel
                if physics.task != "deblurring" and not hasattr(physics, "A_dagger"):
                    raise NotImplementedError("--noise2inverse needs physics.A_dagger outside deblurring")
                wrapped = Noise2InverseModel(backbone=lambda v: model(v.contiguous()), task=physics.task,
                                             physics_filter=getattr(physics, "filter", None),
                                             degradation_inverse_fn=getattr(physics, "A_dagger", None))
                x_hat = wrapped(y)
            elif args.r2r:                                    # :127-134
                x_hat = torch.zeros_like(x)
                for _ in range(args.r2r_itercount):
                    pert = torch.randn_like(y) * physics.noise_model.sigma
                    x_hat += model((y + 0.5 * pert).contiguous())
                x_hat /= args.r2r_itercount
            else:
                x_hat = model(y.contiguous())
        x = quantize_and_clamp(x) if x is not None else None
        y = quantize_and_clamp(y)
        x_hat = quantize_and_clamp(x_hat)
        if x is not None:
            psnr_val, ssim_val, lpips_val = compute_metrics(x.squeeze(0), x_hat.squeeze(0))
            psnr_list.append(psnr_val)
            ssim_list.append(ssim_val)
            lpips_list.append(lpips_val)
            if args.print_all_metrics:
                print(f"METRICS_{i}: PSNR: {psnr_val:.2f}, SSIM: {ssim_val:.4f}, LIPS: {lpips_val:.4f}")
        if args.save_images:
            assert args.out_dir is not None
            entry = basename_table.get(i, f"{i}.png")
            for folder, im in (("ground_truth", x), ("predictors", y), ("estimates", x_hat)):
                if im is None:
                    continue
                path = os.path.join(args.out_dir, folder, entry)
                os.makedirs(dirname(path), exist_ok=True)
                save_image(im, path)

    n = len(psnr_list)
    if n != 0:
        print(f"N: {n}")
        print(f"PSNR: {np.mean(psnr_list):.2f}")
        print(f"PSNR std: {np.std(psnr_list):.2f}")
        print(f"SSIM: {np.mean(ssim_list):.4f}")
        print(f"SSIM std: {np.std(ssim_list):.4f}")
        print(f"LPIPS: {np.mean(lpips_list):.4f}")
        print(f"LPIPS std: {np.std(lpips_list):.4f}")
    return psnr_list


if __name__ == "__main__":
    main()
