"""(x, y) pairs: ground truth + its seeded synthetic measurement (reference: src/datasets/synthetic_dataset.py)."""
from os import environ

from torch.utils.data import Dataset

from .ground_truth import GroundTruthDataset


def homogeneous_measurement(physics_manager, x, y):
    """HOMOGENEOUS_SWINIR (reference :43-53): for super-resolution the low-resolution measurement is brought to x's
    size by plain bicubic interpolation (F.interpolate(y, x.shape[-2:], mode="bicubic", align_corners=False)), so that
    the upscale-1 SwinIR of models/__init__.py maps same-size images; a no-op otherwise."""
    if "HOMOGENEOUS_SWINIR" not in environ or physics_manager.task != "sr":
        return y
    from physics._ops import resample_to_size
    return resample_to_size(y.contiguous(), x.shape[-2], x.shape[-1], antialias=False)


class SyntheticDataset(Dataset):
    def __init__(self, blueprint, device, deterministic_measurements, unique_seeds, physics):
        super().__init__()
        self.device = device
        self.deterministic_measurements = deterministic_measurements
        self.unique_seeds = unique_seeds
        self.physics_manager = getattr(physics, "__manager")
        self.ground_truth_dataset = GroundTruthDataset(blueprint=blueprint, device=device,
                                                       **blueprint[GroundTruthDataset.__name__])

    def __getitem__(self, index):
        x = self.ground_truth_dataset[index].to(self.device)
        if self.deterministic_measurements:
            seed = self.ground_truth_dataset.get_unique_id(index) if self.unique_seeds else 0
        else:
            seed = None
        y = self.physics_manager.randomly_degrade(x.unsqueeze(0).contiguous(), seed=seed).squeeze(0)
        y = homogeneous_measurement(self.physics_manager, x, y)
        return x, y

    def __len__(self):
        return len(self.ground_truth_dataset)
