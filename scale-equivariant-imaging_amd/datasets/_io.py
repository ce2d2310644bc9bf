"""PNG decode without torchvision (reference: torchvision.io.read_image, default mode UNCHANGED, as called by
src/datasets/div2k.py:29 and single_image.py:24): uint8 tensor (C, H, W) with the file's own channel count."""
import numpy as np
import torch


def read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "P":                       # libpng expands palettes; so does torchvision's decoder
            im = im.convert("RGBA" if "transparency" in im.info else "RGB")
        elif im.mode not in ("L", "LA", "RGB", "RGBA"):
            im = im.convert("RGB")
        a = np.asarray(im, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
