"""Test-time transform in front of the engine: resize(256) -> centre-crop(224) -> normalise.

Counterpart of ``build_test_transform(person_crop=False)`` (workoutdetector/datasets/build.py:131-136)
with torchvision-0.13 *tensor* semantics: ``Resize(int)`` sets the short side to ``size`` and the long
side to ``int(size * long / short)``, bilinear, ``align_corners=False``, no antialias; ``CenterCrop``
starts at ``int(round((dim - crop) / 2))``; ``Normalize`` uses the ImageNet mean/std.

Input-scaling quirk: ``inference_dataset`` feeds float32 frames with
values 0..255 (``torch.cat`` promotion, utils/inference_count.py:412-414) so ``ConvertImageDtype`` is a
no-op and frames are NOT divided by 255.  ``scale_255=False`` (default) reproduces that; ``True`` is the
"fixed" behaviour.

Runs as torch ops on whatever device the frames live on (the engine's GPU in the dataset driver).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

INPUT_SIZE = 224
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resized_hw(h: int, w: int, size: int = 256) -> Tuple[int, int]:
    return (size, int(size * w / h)) if h <= w else (int(size * h / w), size)


def crop_offsets(h: int, w: int, crop: int = INPUT_SIZE) -> Tuple[int, int]:
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


class TestTransform:
    """Callable on float tensors [T,3,H,W] -> [T,3,crop,crop]."""

    __test__ = False  # not a pytest class

    def __init__(self, size: int = 256, crop: int = INPUT_SIZE, scale_255: bool = False):
        self.size, self.crop, self.scale_255 = size, crop, scale_255

    def __call__(self, frames_tchw: torch.Tensor) -> torch.Tensor:
        x = frames_tchw.to(torch.float32)
        if self.scale_255:
            x = x / 255.0
        nh, nw = resized_hw(x.shape[-2], x.shape[-1], self.size)
        x = F.interpolate(x, size=(nh, nw), mode='bilinear', align_corners=False)
        top, left = crop_offsets(nh, nw, self.crop)
        x = x[..., top:top + self.crop, left:left + self.crop]
        mean = torch.tensor(MEAN, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(STD, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
        return ((x - mean) / std).contiguous()

    def __repr__(self):
        return (f'TestTransform(Resize({self.size}), CenterCrop({self.crop}), Normalize(ImageNet), '
                f'scale_255={self.scale_255})')


def build_test_transform(person_crop: bool = False, scale_255: bool = False) -> TestTransform:
    if person_crop:
        raise NotImplementedError('person_crop needs the Faster-RCNN detector, which is outside the hot path')
    return TestTransform(scale_255=scale_255)
