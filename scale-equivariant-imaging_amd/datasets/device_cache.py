"""GPU-resident training pairs (at >1000 images/s the reference's per-item path -- decode, resize,
H2D copy, fork_rng + seeded blur of ONE image, crop, collate, with num_workers=0 -- cannot feed the device).

The reference's measurements are deterministic per image id (`--SyntheticDataset__deterministic_measurements`
with unique seeds, src/datasets/synthetic_dataset.py:30-36), so every (x, y) pair can be produced ONCE through
the same dataset objects, kept in HBM (DIV2K at 256 px: 800 x ~2.4 MB), and a step then only draws crop
offsets and gathers the crops on the device. What changes with respect to the DataLoader path: the order in
which the host RNG is consumed (so the sequence of crops differs), nothing else; `css` re-degradation and
non-deterministic measurements need fresh noise per visit and are refused.
"""
import torch

from crop import CropPair


class DeviceResidentPairs:
    def __init__(self, synthetic_dataset, physics, crop_size, crop_location="random", hotfix_sr_crop=False,
                 rank=0, world=1):
        """synthetic_dataset: map-style dataset of full-size (x, y) device tensors (datasets.SyntheticDataset or
        SyntheticPairs); rank/world: this process keeps items rank, rank + world, ... of the index list padded by
        wrapping around to a multiple of `world` (torch's DistributedSampler rule), so every rank holds the SAME
        number of pairs, runs the same number of steps and takes the same launch path on the last, short batch --
        ranks that disagree on either would hang in, or corrupt, the gradient all-reduce."""
        if getattr(synthetic_dataset, "deterministic_measurements", True) is not True:
            raise ValueError("the device cache needs deterministic measurements")
        self.ratio = physics.rate if physics.task == "sr" else 1
        self.crop = CropPair(location="random", size=48) if hotfix_sr_crop else \
            CropPair(location=crop_location, size=crop_size)
        self.pairs = []
        with torch.no_grad():
            n = len(synthetic_dataset)
            padded = [i % n for i in range(-(-n // world) * world)]
            for index in padded[rank::world]:
                x, y = synthetic_dataset[index]
                self.pairs.append((x.contiguous(), y.contiguous()))

    def __len__(self):
        return len(self.pairs)

    def nbytes(self):
        return sum(x.numel() * x.element_size() + y.numel() * y.element_size() for x, y in self.pairs)

    def batches(self, batch_size, shuffle=True, drop_last=False):
        """One epoch: yields (x, y) batches of crops, every cached pair exactly once."""
        n = len(self.pairs)
        order = torch.randperm(n).tolist() if shuffle else list(range(n))
        for start in range(0, n, batch_size):
            idx = order[start:start + batch_size]
            if drop_last and len(idx) < batch_size:
                return
            xs, ys = zip(*(self.crop(*self.pairs[i], xy_size_ratio=self.ratio) for i in idx))
            yield torch.stack(xs), torch.stack(ys)
