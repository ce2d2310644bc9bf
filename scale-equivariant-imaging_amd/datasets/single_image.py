"""One image repeated (reference: src/datasets/single_image.py)."""
import torch
from torch.utils.data import Dataset

from ._io import read_image


class SingleImageDataset(Dataset):
    def __init__(self, image_path, duplicates_count, download=False):
        self.duplicates_count = duplicates_count
        self.image_path = image_path
        self.im = None

    def __len__(self):
        return self.duplicates_count

    def __getitem__(self, idx):
        if self.im is None:
            self.im = read_image(self.image_path).to(torch.float) / 255.0
        return self.im
