"""Ground-truth images at the training resolution (reference: src/datasets/ground_truth.py).

`TF.resize(x, size=256, BICUBIC, antialias=True)` of the reference is torchvision's "shorter edge -> size" rule
followed by F.interpolate(size=..., mode="bicubic", antialias=True): the same separable antialiased-bicubic
filter as the SR physics operator, at the scale in/out of each axis. Here it runs as one launch of the banded
separable resampler (sei_resample_sepband) on the training device; decode stays on the host (PIL)."""
from torch.utils.data import Dataset

from .div2k import Div2K
from .single_image import SingleImageDataset


def resized_hw(h, w, size):
    """torchvision.transforms.functional.resize(size=int): the shorter edge becomes `size`, the longer one
    int(size * long / short); an image whose shorter edge already equals `size` is returned unchanged."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def resize_aa_bicubic(x, out_h, out_w):
    """(C, H, W) or (B, C, H, W) float32 on the GPU -> (.., out_h, out_w)."""
    from physics._ops import resample_to_size
    return resample_to_size(x, out_h, out_w)


class GroundTruthDataset(Dataset):
    def __init__(self, blueprint, datasets_dir, dataset_name, split, download, size, memoize_gt, device=None):
        super().__init__()
        self.size = size
        self.memoize_gt = memoize_gt
        self.device = device
        dataset_name = dataset_name.lower()
        if dataset_name == "div2k":
            self.dataset = Div2K(split, datasets_dir, download=download)
        elif dataset_name == "single_image":
            self.dataset = SingleImageDataset(**blueprint[SingleImageDataset.__name__])
        elif dataset_name in ("urban100", "ct", "fmd"):
            raise NotImplementedError(f"--dataset {dataset_name}: only the div2k and single_image readers are "
                                      "rebuilt (the others differ in file layout only)")
        else:
            raise ValueError(f"Unknown dataset: {dataset_name}")
        self._cache = {}

    def get_unique_id(self, index):
        if hasattr(self.dataset, "get_unique_id"):
            return self.dataset.get_unique_id(index)
        return index

    def _load(self, index):
        x = self.dataset[index]
        if self.size is not None:
            oh, ow = resized_hw(x.shape[-2], x.shape[-1], self.size)
            if (oh, ow) != tuple(x.shape[-2:]):
                dev = self.device if self.device is not None else "cuda"
                x = resize_aa_bicubic(x.to(dev).contiguous(), oh, ow)
        return x

    def __getitem__(self, index):
        if not self.memoize_gt:
            return self._load(index)
        if index not in self._cache:                     # the reference keeps the memoised copy on the host
            x = self._load(index)
            self._cache[index] = (x.device, x.to("cpu"))
        device, x = self._cache[index]
        return x.to(device)

    def __len__(self):
        return len(self.dataset)
