"""Training / evaluation data (reference call surface: src/datasets/__init__.py).

`get_dataset(args, purpose, physics, device, _HOTFIX)` builds, as the reference does,
    GroundTruthDataset (file reader + antialiased resize to 256)  ->  SyntheticDataset (seeded y = noise(A(x)))
    ->  TrainingDataset (random paired crops; `css` re-degrades y)  |  TestDataset (x cropped to fit y)
for `--dataset div2k` and `--dataset single_image` (PNG decode through PIL, the resize through the banded HIP
resampler). `--dataset synthetic` is this build's own file-less stand-in with the same (x, y) contract
(uniform-noise images through the real physics operator): it is what bench.py and the GPU tests use, since no
image data ships with the repository. The `noise2inverse` flag travels as upstream: the training wrapper stores and
ignores it, the test wrapper trims deblurring measurements to even sizes for noise2inverse.py's row slicing.
The HOMOGENEOUS_SWINIR environment switch is honoured on this side too (reference :21-27,35-40,79-82 and
synthetic_dataset.py:43-53): super-resolution measurements are upsampled to x's size with plain bicubic interpolation
and training pairs are same-size 48-pixel crops, matching the upscale-1 SwinIR of models/__init__.py and the un-cropped
Loss of losses/__init__.py. Not rebuilt: urban100 / ct / fmd readers.
"""
from os import environ

import torch
from torch.nn import Module
from torch.utils.data import Dataset as BaseDataset

from crop import CropPair
from .ground_truth import GroundTruthDataset
from .single_image import SingleImageDataset
from .synthetic_dataset import SyntheticDataset, homogeneous_measurement


class SyntheticPairs(BaseDataset):
    """File-less pairs: x = seeded uniform noise in [0,1], y = the physics manager's seeded measurement of it,
    deterministic per index exactly like SyntheticDataset.__getitem__ (synthetic_dataset.py:26-55)."""

    def __init__(self, physics, device, length=800, size=256, seed=1234, hotfix_sr_crop=False, css=False):
        self.physics, self.device, self.length, self.size = physics, device, length, size
        self.seed = seed
        self.hotfix = hotfix_sr_crop
        self.css = css

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed + index)
        x = torch.rand((3, self.size, self.size), generator=g).to(self.device)
        manager = getattr(self.physics, "__manager")
        y = manager.randomly_degrade(x[None], seed=index)[0]
        y = homogeneous_measurement(manager, x, y)
        if self.css:                                        # as TrainingDataset below (reference :70-76)
            x, y = y, manager.randomly_degrade(y[None].contiguous(), seed=None)[0]
        if self.hotfix:
            if "HOMOGENEOUS_SWINIR" in environ:             # same-size pairs: 48 / 48 crops (reference :21-40,79-82)
                return CropPair(location="random", size=48)(x, y, xy_size_ratio=1)
            return CropPair(location="random", size=48)(x, y, xy_size_ratio=self.physics.rate)
        return x, y


class PrepareTrainingPairs(Module):
    """Random (or centre) paired crop of a dataset item (reference :16-47)."""

    def __init__(self, physics, crop_size, crop_location):
        super().__init__()
        self.physics = physics
        self.crop_size = crop_size
        self.crop_location = crop_location
        if "HOMOGENEOUS_SWINIR" in environ:                 # reference :21-27: 48-pixel training crops
            self.crop_size = 48

    def forward(self, x, y):
        ratio = self.physics.rate if self.physics.task == "sr" else 1
        if "HOMOGENEOUS_SWINIR" in environ:                 # reference :35-40: x and y have the same size
            ratio = 1
        return CropPair(location=self.crop_location, size=self.crop_size)(x, y, xy_size_ratio=ratio)


class TrainingDataset(BaseDataset):
    def __init__(self, synthetic_dataset, physics, css, noise2inverse, prepare_training_pairs, _HOTFIX):
        super().__init__()
        self.noise2inverse = noise2inverse                  # stored and never read, as upstream (:63)
        self.synthetic_dataset = synthetic_dataset
        self.physics = physics
        self.css = css
        self.prepare_training_pairs = prepare_training_pairs
        self.important_unnamed_flag = _HOTFIX

    def __getitem__(self, index):
        x, y = self.synthetic_dataset[index]
        if self.css:                                        # the measurement becomes the target (reference :70-76)
            manager = getattr(self.physics, "__manager")
            z = manager.randomly_degrade(y.unsqueeze(0).contiguous(), seed=None).squeeze(0)
            x, y = y, z
        if self.important_unnamed_flag:                     # SR: 48 / 48*rate crops before batching (:78-85)
            if "HOMOGENEOUS_SWINIR" in environ:             # (:79-82: the same-size 48 / 48 crop instead)
                return self.prepare_training_pairs(x, y)
            return CropPair(location="random", size=48)(x, y, xy_size_ratio=self.physics.rate)
        return self.prepare_training_pairs(x, y)

    def __len__(self):
        return len(self.synthetic_dataset)


class TestDataset(BaseDataset):
    def __init__(self, synthetic_dataset, noise2inverse, physics):
        super().__init__()
        self.noise2inverse = noise2inverse
        self.synthetic_dataset = synthetic_dataset
        self.physics = physics

    def __getitem__(self, index):
        x, y = self.synthetic_dataset[index]
        if self.noise2inverse and self.physics.task == "deblurring":     # even height and width (reference :112-118)
            y = y[:, :2 * (y.shape[1] // 2), :2 * (y.shape[2] // 2)]
        if x.shape != y.shape:                              # crop x to a multiple of y's size (reference :121-128)
            h, w = y.shape[1], y.shape[2]
            f = self.physics.rate if self.physics.task == "sr" else 1
            x = x[:, :h * f, :w * f]
        return x, y

    def __len__(self):
        return len(self.synthetic_dataset)


class Dataset(BaseDataset):
    def __init__(self, blueprint, purpose, physics, css, noise2inverse, device, _HOTFIX):
        super().__init__()
        synthetic_dataset = SyntheticDataset(blueprint=blueprint, device=device, physics=physics,
                                             **blueprint[SyntheticDataset.__name__])
        if purpose == "train":
            prepare = PrepareTrainingPairs(physics=physics, **blueprint[PrepareTrainingPairs.__name__])
            self.dataset = TrainingDataset(synthetic_dataset=synthetic_dataset, physics=physics, css=css,
                                           noise2inverse=noise2inverse, prepare_training_pairs=prepare,
                                           _HOTFIX=_HOTFIX)
        elif purpose == "test":
            self.dataset = TestDataset(synthetic_dataset=synthetic_dataset, noise2inverse=noise2inverse,
                                       physics=physics)
        else:
            raise ValueError(f"Unknown purpose: {purpose}")

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        return self.dataset[index]


def get_dataset(args, purpose, physics, device, _HOTFIX=False):
    if purpose == "test":
        noise2inverse, css = getattr(args, "noise2inverse", False), False
    elif purpose == "train":
        noise2inverse, css = args.method == "noise2inverse", args.method == "css"
    else:
        raise ValueError(f"Unknown purpose: {purpose}")
    if args.dataset == "synthetic":                         # this build's file-less stand-in
        return SyntheticPairs(physics, device, hotfix_sr_crop=_HOTFIX and purpose == "train", css=css)
    blueprint = {
        GroundTruthDataset.__name__: {
            "dataset_name": args.dataset,
            "datasets_dir": args.GroundTruthDataset__datasets_dir,
            "download": args.GroundTruthDataset__download,
            "size": args.GroundTruthDataset__size,
            "split": args.GroundTruthDataset__split,
            "memoize_gt": args.memoize_gt,
        },
        PrepareTrainingPairs.__name__: {
            "crop_size": args.PrepareTrainingPairs__crop_size,
            "crop_location": args.PrepareTrainingPairs__crop_location,
        },
        SingleImageDataset.__name__: {
            "image_path": args.SingleImageDataset__image_path,
            "duplicates_count": args.SingleImageDataset__duplicates_count,
        },
        SyntheticDataset.__name__: {
            "unique_seeds": args.SyntheticDataset__unique_seeds,
            "deterministic_measurements": args.SyntheticDataset__deterministic_measurements,
        },
    }
    return Dataset(blueprint=blueprint, device=device, physics=physics, purpose=purpose, css=css,
                   noise2inverse=noise2inverse, _HOTFIX=_HOTFIX)
