"""Dataset glue of the hot path (PKG/data/loader.py:23-293, 390-420), re-designed so a batch is
BORN on the GPU: synthetic frames, the per-sample weather choice, the corruption kernels and the
fused Normalize+ToTensorV2 all run on device; only the scalar draws happen on the host.

File discovery / decoding / resizing of real Cityscapes-KITTI trees is host I/O and out of scope
(SURVEY §2 rows 3, 5); like the reference when no data is found (loader.py:103-105), the dataset
serves synthetic samples: 100 for 'train', 20 otherwise (:165-179), uint8 images uniform on
0..254 (:206) and uint8 labels uniform on 0..18 (:231).
"""
from __future__ import annotations

import logging
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .preprocessing import DepthEstimationPreprocessor, WeatherDegradationTransforms

logger = logging.getLogger(__name__)


class CityscapesKITTIDataset:
    """Batch source with the reference dataset's constructor arguments.  Iterate it through
    `create_dataloader` (or `.batches()`): each batch is the reference's collated dict
    {'image' f32[B,3,H,W], 'label' u8[B,H,W], 'weather_condition' list[str], 'dataset' list[str],
    'depth' f32[B,H,W] when include_depth}
    on the GPU."""

    def __init__(self, data_root: str = "data", split: str = "train", image_size: Tuple[int, int] = (512, 1024),
                 weather_conditions: Optional[List[str]] = None, apply_augmentation: bool = True, include_depth: bool = True,
                 dataset_type: str = "combined", device="cuda", rng: str = "philox", weather_schedule: str = "random",
                 num_samples: Optional[int] = None, seed: int = 42) -> None:
        self.data_root, self.split, self.image_size = data_root, split, tuple(image_size)
        self.weather_conditions = weather_conditions or ["clean", "fog", "rain", "snow", "night"]
        self.apply_augmentation, self.include_depth, self.dataset_type = apply_augmentation, include_depth, dataset_type
        self.device = torch.device(device)
        self.weather_schedule = weather_schedule          # 'random' (loader.py:265) or 'round_robin' (bench)
        self.num_samples = num_samples if num_samples is not None else (100 if split == "train" else 20)
        self.seed = int(seed)
        # The reference redraws pixels, the weather choice and every corruption parameter on each __getitem__
        # (loader.py:206, 231, 265), so a training epoch never repeats.  Here every draw is a function of
        # (seed, split, epoch, index): rank-independent, fresh per training epoch (the loader bumps `epoch` on every
        # pass over a 'train' split), frozen for val / test so a validation curve compares like with like.
        self.epoch = 0
        self.weather_transforms = WeatherDegradationTransforms(rng=rng, device=self.device)
        self.weather_transforms._frame_seed = self._frame_seed()   # per-frame streams keyed by (seed, epoch, global index); no global reseed
        self.depth_preprocessor = DepthEstimationPreprocessor(self.device) if include_depth else None   # loader.py:68-69
        self._gen = torch.Generator(device=self.device)
        logger.info("Generated %d synthetic samples for testing", self.num_samples)

    def __len__(self) -> int:
        return self.num_samples

    def _epoch_term(self) -> int:
        return int(self.epoch) if self.split == "train" else 0

    def _frame_seed(self) -> int:
        return (self.seed + 0x632BE5AB * self._epoch_term()) & 0x7FFFFFFF

    def set_epoch(self, epoch: int) -> None:
        """Select the draw stream of a training epoch (no effect on val / test splits)."""
        self.epoch = int(epoch)
        self.weather_transforms._frame_seed = self._frame_seed()

    def _sample_key(self, index: int) -> int:
        split_salt = {"train": 0, "val": 1, "test": 2}.get(self.split, 3)
        return (self.seed * 1000003 + split_salt * 7919 + self._epoch_term() * 0x2545F491 + int(index) * 0x9E3779B1) & 0x7FFFFFFFFFFFFFFF

    def synth_raw(self, start: int, n: int):
        """uint8 frames / labels generated on device (shapes and ranges of loader.py:206, 231).  Sample i
        is a function of (seed, split, i) only, so every rank of a sharded run — and the single-process
        run — sees the same sample under the same global index."""
        h, w = self.image_size
        imgs = torch.empty(n, h, w, 3, dtype=torch.uint8, device=self.device)
        labels = torch.empty(n, h, w, dtype=torch.uint8, device=self.device)
        for k in range(n):
            self._gen.manual_seed(self._sample_key(start + k))
            imgs[k] = torch.randint(0, 255, (h, w, 3), dtype=torch.uint8, device=self.device, generator=self._gen)
            labels[k] = torch.randint(0, 19, (h, w), dtype=torch.uint8, device=self.device, generator=self._gen)
        return imgs, labels

    def choose_conditions(self, start: int, n: int) -> List[str]:
        if self.weather_schedule == "round_robin":
            return [self.weather_conditions[(start + i) % len(self.weather_conditions)] for i in range(n)]
        # loader.py:265 draws per __getitem__; here the draw is keyed by the global sample index
        return [str(np.random.RandomState(self._sample_key(start + i) & 0xFFFFFFFF).choice(self.weather_conditions)) for i in range(n)]

    def make_batch(self, start: int, n: int, raw=None) -> Dict[str, object]:
        imgs, labels = raw if raw is not None else self.synth_raw(start, n)
        conds = self.choose_conditions(start, n)
        ids = list(range(start, start + n)) if self.weather_transforms.rng == "philox" else None   # numpy mode keeps the reference's global stream
        h, w = self.image_size
        image = torch.empty(n, 3, h, w, dtype=torch.float32, device=self.device)
        batch = {"image": image, "label": labels, "weather_condition": conds, "dataset": ["synthetic"] * n}
        if self.depth_preprocessor is None:
            self.weather_transforms.apply_batch(imgs, conds, norm_out=image, frame_ids=ids)
        else:
            # the depth target is estimated from the CORRUPTED uint8 frame (loader.py:264-272), so
            # the transforms also write their uint8 output
            frames = torch.empty_like(imgs)
            self.weather_transforms.apply_batch(imgs, conds, out=frames, norm_out=image, frame_ids=ids)
            batch["depth"] = self.depth_preprocessor.estimate_depth_batch(frames)
        return batch

    def batches(self, batch_size: int, drop_last: bool = False, rank: int = 0, world_size: int = 1) -> Iterator[Dict[str, object]]:
        """Contiguous block sharding of the sample index range over ranks."""
        per = (self.num_samples + world_size - 1) // world_size
        lo, hi = min(rank * per, self.num_samples), min((rank + 1) * per, self.num_samples)
        i = lo
        while i < hi:
            n = min(batch_size, hi - i)
            if n < batch_size and drop_last:
                break
            yield self.make_batch(i, n)
            i += n


# (alpha, beta, channel-2 gain) of _apply_style_transfer, PKG/data/loader.py:372-385
STYLE_PARAMS = {"fog": (0.8, 30.0, None), "rain": (1.2, -10.0, 1.1), "snow": (0.9, 20.0, None), "night": (0.4, -20.0, 1.3)}


def style_lut(weather_type: str) -> np.ndarray:
    """uint8 [3,256] table of `_apply_style_transfer` for one weather type: cv2.convertScaleAbs on
    8-bit data is saturate_cast<uchar>(|v*(float)alpha + (float)beta|) (float32 arithmetic, round
    half to even), then `image[:,:,2] = np.clip(image[:,:,2]*gain, 0, 255)` is a float64 product
    truncated by the uint8 store.  Identity for types the reference does not style."""
    v = np.arange(256, dtype=np.float32)
    if weather_type not in STYLE_PARAMS:
        return np.tile(np.arange(256, dtype=np.uint8), (3, 1))
    alpha, beta, gain = STYLE_PARAMS[weather_type]
    base = np.clip(np.rint(np.abs(v * np.float32(alpha) + np.float32(beta))), 0, 255).astype(np.uint8)
    lut = np.tile(base, (3, 1))
    if gain is not None:
        lut[2] = np.clip(base * gain, 0, 255).astype(np.uint8)
    return lut


class WeatherAugmentationPipeline:
    """PKG/data/loader.py:296-387: fixed-intensity weather followed, with probability
    `style_transfer_prob`, by the colour 'style transfer'.  Frames may be numpy HWC uint8 (reference
    convention, returned as numpy) or uint8 device tensors [H,W,3] / [B,H,W,3]."""

    def __init__(self, weather_intensities: Optional[Dict[str, float]] = None, style_transfer_prob: float = 0.3,
                 device="cuda", rng: str = "numpy", **kwargs) -> None:
        self.weather_intensities = weather_intensities or {"fog": 0.7, "rain": 0.5, "snow": 0.6, "night": 0.8}
        self.style_transfer_prob = style_transfer_prob
        self.device = torch.device(device)
        self.weather_transforms = WeatherDegradationTransforms(rng=rng, device=self.device)
        self._types = list(STYLE_PARAMS)
        self._luts = None

    def _device_luts(self) -> torch.Tensor:
        if self._luts is None:
            self._luts = torch.from_numpy(np.stack([style_lut(t) for t in self._types])).to(self.device)
        return self._luts

    def apply_domain_adaptation_augmentation(self, image, target_weather: Optional[str] = None):
        if target_weather is None:
            target_weather = str(np.random.choice(list(self.weather_intensities.keys())))          # :347
        out = self.weather_transforms.apply_weather_effect(image, target_weather, intensity=self.weather_intensities[target_weather])
        if np.random.random() < self.style_transfer_prob:                                           # :355
            out = self._apply_style_transfer(out, target_weather)
        return out

    def _apply_style_transfer(self, image, weather_type: str):
        if weather_type not in STYLE_PARAMS:
            return image
        is_np = isinstance(image, np.ndarray)
        dev = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(self.device) if is_np else image
        batch = dev.unsqueeze(0) if dev.dim() == 3 else dev
        which = torch.full((batch.shape[0],), self._types.index(weather_type), dtype=torch.int32, device=self.device)
        out = ops.lut3_apply(batch, self._device_luts(), which)
        out = out[0] if dev.dim() == 3 else out
        return out.cpu().numpy() if is_np else out


class _Loader:
    def __init__(self, dataset, batch_size, drop_last, rank, world_size):
        self.dataset, self.batch_size, self.drop_last, self.rank, self.world_size = dataset, batch_size, drop_last, rank, world_size
        self._passes = 0

    def __iter__(self):
        it = self.dataset.batches(self.batch_size, self.drop_last, self.rank, self.world_size)
        if hasattr(self.dataset, "set_epoch"):
            # generators run lazily: fix this pass's epoch now, hand the NEXT pass the next one (every rank iterates its
            # loader once per epoch, so the counters agree across ranks without communication)
            first = self._passes
            self._passes += 1

            def run():
                self.dataset.set_epoch(first)
                yield from it
            return run()
        return it

    def __len__(self):
        per = (len(self.dataset) + self.world_size - 1) // self.world_size
        n = max(0, min((self.rank + 1) * per, len(self.dataset)) - self.rank * per)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


def create_dataloader(dataset, batch_size: int = 8, shuffle: bool = True, num_workers: int = 4, pin_memory: bool = True,
                      rank: int = 0, world_size: int = 1):
    """PKG/data/loader.py:390-420: `drop_last = shuffle`.  Worker processes / pinned memory are
    meaningless for a device-resident source and are accepted and ignored."""
    return _Loader(dataset, batch_size, drop_last=shuffle, rank=rank, world_size=world_size)
