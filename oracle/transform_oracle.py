"""CPU restatement of the pre-step of the hot path: clip windows + the test transform.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, in /root/reference:
  build_test_transform(person_crop=False)   workoutdetector/datasets/build.py:131-136
      ConvertImageDtype(float32) -> Resize(256) -> CenterCrop(224) -> Normalize(ImageNet)
  clip construction                         workoutdetector/utils/inference_count.py:411-414
      vid[i:i+16:2], zero-pad to 8 frames with a *float32* zeros tensor; torch.cat promotes
      the uint8 clip to float32 0..255, so ConvertImageDtype is a no-op and frames are never
      divided by 255.  ``scale_255=True`` is the "fixed" variant.
torchvision 0.13 tensor semantics (package absent; restated from its public definition):
  Resize(int): short side -> size, long side -> int(size * long / short); bilinear,
  align_corners=False, no antialias for tensors.  CenterCrop: top = int(round((H - ch) / 2)).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resized_hw(h: int, w: int, size: int = 256) -> Tuple[int, int]:
    if h <= w:
        return size, int(size * w / h)
    return int(size * h / w), size


def crop_offsets(h: int, w: int, crop: int = 224) -> Tuple[int, int]:
    return int(round((h - crop) / 2.0)), int(round((w - crop) / 2.0))


def test_transform(frames_tchw: torch.Tensor, size: int = 256, crop: int = 224,
                   scale_255: bool = False) -> torch.Tensor:
    """[T,3,H,W] float32 (values 0..255 on the reference path) -> [T,3,crop,crop] float32."""
    x = frames_tchw.to(torch.float32)
    if scale_255:
        x = x / 255.0
    nh, nw = resized_hw(x.shape[-2], x.shape[-1], size)
    x = F.interpolate(x, size=(nh, nw), mode='bilinear', align_corners=False)
    top, left = crop_offsets(nh, nw, crop)
    x = x[..., top:top + crop, left:left + crop]
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - mean) / std


def make_clip(video_thwc_u8: torch.Tensor, start: int) -> torch.Tensor:
    """One sparse-sampled window: 8 frames spanning 16 source frames, tail zero-padded.
    Returns float32 [8,H,W,3] with values 0..255 (the reference's promotion quirk)."""
    clip = video_thwc_u8[start:start + 16:2]
    pad = 8 - clip.shape[0]
    clip = clip.to(torch.float32)
    if pad > 0:
        clip = torch.cat([clip, torch.zeros((pad,) + tuple(clip.shape[1:]), dtype=torch.float32)])
    return clip


def clip_to_input(clip_thwc: torch.Tensor, scale_255: bool = False) -> torch.Tensor:
    """[8,H,W,3] -> network input [1,8,3,224,224] (inference_video, inference_count.py:269-272)."""
    x = clip_thwc.permute(0, 3, 1, 2)
    return test_transform(x, scale_255=scale_255).unsqueeze(0)
