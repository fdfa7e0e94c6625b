"""Host side of the device input pipeline: the transform the reference builds with Dassl's `build_transform`
(INPUT.TRANSFORMS = random_resized_crop / random_flip / normalize, bicubic, CLIP mean/std:
configs/trainers/MVLPT/vit_b16.yaml:8-13) and the ELEVATER eval transform (Resize(BICUBIC) [+ CenterCrop] + ToTensor +
Normalize: trainers/vision_benchmark/evaluation/feature.py:538-553), re-cut for a GPU: the host only draws the random
parameters and packs decoded uint8 images into one pinned buffer; crop, antialiased bicubic resample, flip, ToTensor
and Normalize run in `mvlpt_preprocess` (csrc/preprocess.hip), bit-identical to Pillow + torch CPU.

The parameter sampling follows torchvision's documented algorithms (`RandomResizedCrop.get_params`, `Resize` with an
int size, `CenterCrop`); torchvision is not installed in the build image, so those few integer formulas are restated
from its documentation and are NOT pinned against it — the pixel arithmetic (Pillow) is, by tests/golden/preprocess.npz."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def random_resized_crop_params(height: int, width: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0),
                               generator: Optional[torch.Generator] = None) -> Tuple[int, int, int, int]:
    """(top, left, h, w) as torchvision.transforms.RandomResizedCrop.get_params: 10 attempts at a box of random area
    fraction and log-uniform aspect ratio, then a centre crop clamped to the ratio range."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * float(torch.empty(1).uniform_(scale[0], scale[1], generator=generator))
        aspect = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator)))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, (1,), generator=generator))
            j = int(torch.randint(0, width - w + 1, (1,), generator=generator))
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def resize_shorter_side(height: int, width: int, size: int) -> Tuple[int, int]:
    """Output (h, w) of torchvision `Resize(int)`: shorter side -> size, longer side int(size * long / short)."""
    short, long = (width, height) if width <= height else (height, width)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if width <= height else (new_short, new_long)


def center_crop_offsets(height: int, width: int, out_h: int, out_w: int) -> Tuple[int, int]:
    """(top, left) of torchvision `CenterCrop` (Python round = round-half-even)."""
    if out_h > height or out_w > width:
        raise ValueError("CenterCrop larger than the resized image (padding) is not supported")
    return int(round((height - out_h) / 2.0)), int(round((width - out_w) / 2.0))


class DeviceTransform:
    """train=True : random_resized_crop + random_flip + normalize   (vit_b16.yaml:8-13)
    train=False: Resize(size) + CenterCrop(size) + normalize when `center_crop`, else Resize((size, size)) + normalize
                 (feature.py:538-553).
    Call with a list of decoded uint8 HWC numpy arrays (or uint8 tensors); returns the [B,3,size,size] device batch."""

    def __init__(self, engine, size: int = 224, train: bool = True, center_crop: bool = True, mean: Sequence[float] = CLIP_MEAN,
                 std: Sequence[float] = CLIP_STD, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), flip_p: float = 0.5,
                 out_dtype=torch.float32, generator: Optional[torch.Generator] = None):
        self.engine, self.size, self.train, self.center_crop = engine, int(size), train, center_crop
        self.mean, self.std, self.scale, self.ratio, self.flip_p = tuple(mean), tuple(std), scale, ratio, flip_p
        self.out_dtype, self.generator = out_dtype, generator
        # ring of pinned staging buffers, each guarded by the event of its last H2D copy: the train step does not
        # synchronise per step, so the copy of batch i may still be queued when batch i+1 is packed
        self._ring: List[list] = [[None, None] for _ in range(3)]      # [pinned tensor, torch.cuda.Event]
        self._slot = -1

    def describe(self, shapes: Sequence[Tuple[int, int]]):
        """Descriptors (ctypes array) for images of the given (height, width); draws the random parameters."""
        R = self.size
        descs = (_lib.MvlptImageDesc * len(shapes))()
        off = 0
        for d, (h, w) in zip(descs, shapes):
            d.offset, d.height, d.width = off, h, w
            off += h * w * 3
            if self.train:
                d.crop_top, d.crop_left, d.crop_height, d.crop_width = random_resized_crop_params(h, w, self.scale, self.ratio, self.generator)
                d.resize_height = d.resize_width = R
                d.out_top = d.out_left = 0
                d.flip = int(float(torch.rand(1, generator=self.generator)) < self.flip_p)
            else:
                d.crop_top = d.crop_left = 0
                d.crop_height, d.crop_width = h, w
                if self.center_crop:
                    d.resize_height, d.resize_width = resize_shorter_side(h, w, R)
                    d.out_top, d.out_left = center_crop_offsets(d.resize_height, d.resize_width, R, R)
                else:
                    d.resize_height = d.resize_width = R
                    d.out_top = d.out_left = 0
                d.flip = 0
        return descs, off

    def pack(self, images: Sequence) -> Tuple[torch.Tensor, int]:
        """Copies the decoded images back to back into a (re-used) pinned host buffer."""
        total = sum(int(np.prod(im.shape)) for im in images)
        self._slot = (self._slot + 1) % len(self._ring)
        ent = self._ring[self._slot]
        if ent[1] is not None:
            ent[1].synchronize()                      # the upload that last read this buffer has completed
        if ent[0] is None or ent[0].numel() < total:
            ent[0] = torch.empty(max(total, 1), dtype=torch.uint8)
            if torch.cuda.is_available():
                ent[0] = ent[0].pin_memory()
        pinned = ent[0]
        off = 0
        for im in images:
            a = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
            if a.dtype != torch.uint8 or a.dim() != 3 or a.shape[2] != 3:
                raise ValueError("images must be uint8 HWC with 3 channels")
            n = a.numel()
            pinned[off:off + n].copy_(a.reshape(-1))
            off += n
        return pinned, total

    def __call__(self, images: Sequence, want_u8: bool = False):
        descs, total = self.describe([(int(im.shape[0]), int(im.shape[1])) for im in images])
        host, n = self.pack(images)
        src = host[:n].to(self.engine.device if hasattr(self.engine, "device") else "cuda", non_blocking=True)
        if src.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(src.device))
            self._ring[self._slot][1] = ev
        return self.engine.preprocess(src, descs, self.size, self.mean, self.std, self.out_dtype, want_u8)
