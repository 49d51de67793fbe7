"""
The image pipeline of the reference's VG dataset (dataloaders/visual_genome.py:94-99): SquarePad -> Resize(IM_SCALE)
-> ToTensor -> Normalize, on PIL + torch only (torchvision is not part of this environment).  SquarePad is the
reference's own transform (dataloaders/image_transforms.py:8-13); the other three restate torchvision 0.2's
Resize(int) (bilinear, smaller edge -> size), ToTensor (uint8 HWC -> float CHW / 255) and Normalize.
"""
import numpy as np
import torch
from PIL import Image, ImageOps


class SquarePad(object):
    def __call__(self, img):
        w, h = img.size
        return ImageOps.expand(img, border=(0, 0, max(h - w, 0), max(w - h, 0)),
                               fill=(int(0.485 * 256), int(0.456 * 256), int(0.406 * 256)))


class Resize(object):
    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        w, h = img.size
        if (w <= h and w == self.size) or (h <= w and h == self.size):
            return img
        if w < h:
            ow, oh = self.size, int(self.size * h / w)
        else:
            oh, ow = self.size, int(self.size * w / h)
        return img.resize((ow, oh), self.interpolation)


class ToTensor(object):
    def __call__(self, img):
        arr = np.asarray(img.convert('RGB'), dtype=np.uint8)
        return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
