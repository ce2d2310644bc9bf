"""Training data for the hot path.

The reference's loaders (src/datasets/: PNG decode, resize to 256, seeded synthetic measurement, random
256-crop) are the "next" row N1 of the scope table and are not rebuilt yet. What the training step
needs from them is their OUTPUT contract, which `SyntheticPairs` provides without files: a map-style
dataset of (x, y) pairs with x a 256x256 ground-truth image in [0,1] and y = noise(A(x)) produced by the
physics manager with a per-item seed -- deterministic per index, exactly like
SyntheticDataset.__getitem__ (src/datasets/synthetic_dataset.py:26-55) -- then cropped as
TrainingDataset does (src/datasets/__init__.py:67-90; for SR the 48 / 48*rate "_HOTFIX" crop).
"""
import torch
from torch.utils.data import Dataset

from crop import CropPair


class SyntheticPairs(Dataset):
    def __init__(self, physics, device, length=800, size=256, seed=1234, hotfix_sr_crop=False):
        self.physics, self.device, self.length, self.size = physics, device, length, size
        self.seed = seed
        self.hotfix = hotfix_sr_crop

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed + index)
        x = torch.rand((3, self.size, self.size), generator=g).to(self.device)
        manager = getattr(self.physics, "__manager")
        y = manager.randomly_degrade(x[None], seed=index)[0]
        if self.hotfix:
            return CropPair(location="random", size=48)(x, y, xy_size_ratio=self.physics.rate)
        return x, y


def get_dataset(args, purpose, physics, device, _HOTFIX=False):
    if args.dataset != "synthetic":
        raise NotImplementedError(
            f"--dataset {args.dataset}: the image-file loaders are the next row (N1) of this build's scope; "
            "use --dataset synthetic (uniform-noise images through the real physics operator)")
    return SyntheticPairs(physics, device, hotfix_sr_crop=_HOTFIX)
