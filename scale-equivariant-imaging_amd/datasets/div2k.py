"""DIV2K ground-truth images (reference: src/datasets/div2k.py). No download here (no network): the archive
must already be unpacked under <datasets_dir>/DIV2K/DIV2K_{train,valid}_HR."""
import os

import torch
from torch.utils.data import Dataset

from ._io import read_image


class Div2K(Dataset):
    def __init__(self, split, datasets_dir, download=False):
        super().__init__()
        self.datasets_dir = datasets_dir
        assert split in ["train", "val"]
        self.split = split
        if split == "train":
            self.split_root, self.split_offset, self.split_size = f"{datasets_dir}/DIV2K/DIV2K_train_HR", 1, 800
        else:
            self.split_root, self.split_offset, self.split_size = f"{datasets_dir}/DIV2K/DIV2K_valid_HR", 801, 100
        if download:
            raise NotImplementedError("--download: this build has no network access; unpack DIV2K under "
                                      f"{datasets_dir}/DIV2K yourself")
        if not os.path.isdir(self.split_root):
            raise FileNotFoundError(f"{self.split_root} does not exist (DIV2K {split} split)")

    def __getitem__(self, index):
        index = self.split_offset + index
        x = read_image(f"{self.split_root}/{index:04d}.png")
        return x.to(torch.float) / 255.0

    def __len__(self):
        return self.split_size

    def get_unique_id(self, index):
        return self.split_offset + index - 1
