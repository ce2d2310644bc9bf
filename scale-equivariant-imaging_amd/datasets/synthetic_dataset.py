"""(x, y) pairs: ground truth + its seeded synthetic measurement (reference: src/datasets/synthetic_dataset.py)."""
from torch.utils.data import Dataset

from .ground_truth import GroundTruthDataset


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
        return x, y

    def __len__(self):
        return len(self.ground_truth_dataset)
