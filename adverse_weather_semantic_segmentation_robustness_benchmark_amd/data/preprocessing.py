"""Weather degradation transforms with the reference's class / method names
(PKG/data/preprocessing.py:15-288), executed by the HIP kernels of csrc/weather.hip.

Randomness.  The reference draws everything from numpy's global legacy RNG
in a fixed call order.  Two modes:

* ``rng="numpy"`` (default for the drop-in API): the host draws the SAME numbers in the SAME
  order from ``np.random`` and uploads them, so for a given ``np.random.seed`` the uint8 result
  is what the reference produces (bit-exact for fog / night; rain / snow depend on OpenCV's
  rasteriser, parity unpinned).
* ``rng="philox"``: scalar parameters still come from the host, per-pixel noise is generated
  in-kernel by Philox4x32-10 — nothing per-pixel crosses PCIe.  Same distributions, different
  stream: parity is not defined in this mode; it is the throughput mode the batch path uses.

Inputs may be numpy HWC uint8 arrays (reference calling convention: uploaded, processed,
downloaded) or CUDA uint8 tensors [H,W,3] / [B,H,W,3] (stay on device).
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from .. import ops

logger = logging.getLogger(__name__)

WEATHER_TYPES = ("clean", "fog", "rain", "snow", "night")


# --- host-side parameter draws, in the reference's RNG call order ---------------------------
def draw_fog(h: int, w: int, intensity=None, per_pixel: bool = True, rs=np.random):
    """depth noise first (preprocessing.py:239 via :104), then intensity (:108).  `rs`: the numpy
    global legacy RNG (the reference's stream) or a RandomState with the same methods."""
    noise = rs.normal(0, 10, (h, w)) if per_pixel else None
    if intensity is None:
        intensity = rs.uniform(0.3, 0.9)
    return noise, intensity


def draw_night(h: int, w: int, intensity=None, per_pixel: bool = True, rs=np.random):
    """intensity (:207), brightness factor (:212), noise (:222)."""
    if intensity is None:
        intensity = rs.uniform(0.4, 0.8)
    brightness = 1 - intensity * rs.uniform(0.2, 0.6)
    noise = rs.normal(0, 5.0 / 255.0, (h, w, 3)) if per_pixel else None
    return intensity, brightness, noise


def draw_rain(h: int, w: int, intensity=None, rs=np.random):
    """intensity (:128); per drop x, y, length, thickness in {1,3}, angle (:144-148); end point
    truncated toward zero and clipped into the image (:151-156)."""
    if intensity is None:
        intensity = rs.uniform(0.2, 0.8)
    n = int(100 + intensity * (500 - 100))
    drops = np.empty((n, 5), dtype=np.int32)
    if rs is not np.random:
        # per-frame stream (throughput mode): the same distributions drawn as arrays — no parity with the reference's
        # call-by-call order is defined in this mode, and 5 Python RNG calls per drop are ~8 ms of host time per frame
        x, y = rs.randint(0, w, n), rs.randint(0, h, n)
        length, thick, angle = rs.randint(5, 20, n), rs.choice((1, 3), n), rs.uniform(-15, 15, n)
        drops[:, 0], drops[:, 1] = x, y
        drops[:, 2] = np.clip((x + length * np.sin(np.radians(angle))).astype(np.int64), 0, w - 1)   # int(): truncation toward zero
        drops[:, 3] = np.clip((y + length * np.cos(np.radians(angle))).astype(np.int64), 0, h - 1)
        drops[:, 4] = thick
        return intensity, drops
    for i in range(n):
        x = rs.randint(0, w)
        y = rs.randint(0, h)
        length = rs.randint(5, 20)
        thickness = rs.choice((1, 3))
        angle = rs.uniform(-15, 15)
        ex = np.clip(int(x + length * np.sin(np.radians(angle))), 0, w - 1)
        ey = np.clip(int(y + length * np.cos(np.radians(angle))), 0, h - 1)
        drops[i] = (x, y, ex, ey, thickness)
    return intensity, drops


def draw_snow(h: int, w: int, intensity=None, rs=np.random):
    """intensity (:173); per flake x, y, radius in {2,8} (:189-191); blur kernel in {3,7} (:197)."""
    if intensity is None:
        intensity = rs.uniform(0.2, 0.7)
    n = int(50 + intensity * (200 - 50))
    flakes = np.empty((n, 3), dtype=np.int32)
    if rs is not np.random:                                         # per-frame stream: array draws (see draw_rain)
        flakes[:, 0], flakes[:, 1], flakes[:, 2] = rs.randint(0, w, n), rs.randint(0, h, n), rs.choice((2, 8), n)
    else:
        for i in range(n):
            flakes[i] = (rs.randint(0, w), rs.randint(0, h), rs.choice((2, 8)))
    k = int(rs.choice((3, 7)))
    return intensity, flakes, (k + 1 if k % 2 == 0 else k)


class WeatherDegradationTransforms:
    """PKG/data/preprocessing.py:15-288."""

    def __init__(self, seed: Optional[int] = None, rng: str = "numpy", device: Union[str, torch.device] = "cuda") -> None:
        if seed is not None:
            np.random.seed(seed)
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.rng = rng
        self.device = torch.device(device)
        self._philox_seed = 0x5EED if seed is None else int(seed)
        self._frame_seed = 0x5EED if seed is None else int(seed)      # base of the per-frame streams (apply_batch frame_ids)
        self.fog_parameters = {"beta_range": (0.005, 0.05), "A_range": (0.7, 1.0), "depth_scale": 100.0}
        self.rain_parameters = {"intensity_range": (0.1, 0.8), "drop_size_range": (1, 3), "angle_range": (-15, 15),
                                "num_drops_range": (100, 500)}
        self.snow_parameters = {"intensity_range": (0.1, 0.7), "flake_size_range": (2, 8), "num_flakes_range": (50, 200),
                                "blur_kernel": (3, 7)}
        self.night_parameters = {"brightness_reduction": (0.2, 0.6), "color_shift": {"r": 0.8, "g": 0.85, "b": 1.2},
                                 "noise_std": 5.0}

    def _next_seed(self) -> int:
        self._philox_seed = (self._philox_seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._philox_seed

    # ---- single-image API (reference calling convention) -------------------------------------
    def apply_weather_effect(self, image, weather_type: str, intensity: Optional[float] = None):
        if weather_type == "clean":
            return image                                                             # :78-79
        if weather_type not in WEATHER_TYPES:
            raise ValueError(f"Unknown weather type: {weather_type}")                # :92
        is_np = isinstance(image, np.ndarray)
        dev_img = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(self.device) if is_np else image
        batch = dev_img.unsqueeze(0) if dev_img.dim() == 3 else dev_img
        out = self.apply_batch(batch, [weather_type] * batch.shape[0], intensities=[intensity] * batch.shape[0])
        out = out[0] if dev_img.dim() == 3 else out
        return out.cpu().numpy() if is_np else out

    # ---- batched device API (one launch per condition present in the batch) -----------------
    def frame_stream(self, frame_id: int):
        """(RandomState, Philox seed) of one GLOBAL sample index: every scalar draw and every per-pixel
        noise value of that frame is a function of (seed, frame_id) only — not of the batch it lands in,
        the rank that owns it or the order of earlier calls — so a sharded run reproduces the
        single-process run frame for frame (pooled mIoU identical at any GPU count)."""
        key = (self._frame_seed * 0x9E3779B97F4A7C15 + (int(frame_id) + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        return np.random.RandomState([key & 0xFFFFFFFF, key >> 32]), key

    def apply_batch(self, imgs: torch.Tensor, conditions: Sequence[str], intensities: Optional[Sequence] = None,
                    out: Optional[torch.Tensor] = None, norm_out: Optional[torch.Tensor] = None,
                    frame_ids: Optional[Sequence[int]] = None) -> Optional[torch.Tensor]:
        """imgs uint8 [B,H,W,3] on device.  Writes the corrupted uint8 frames to `out` (allocated
        when both outputs are None) and/or the normalised float32 [B,3,H,W] tensor to `norm_out`
        (the loader's Normalize+ToTensorV2, fused).  'clean' frames are copied / normalised.
        `frame_ids` (global sample indices): draw each frame's randomness from frame_stream(id)
        instead of the shared sequential stream."""
        B, H, W, _ = imgs.shape
        if intensities is None:
            intensities = [None] * B
        if out is None and norm_out is None:
            out = torch.empty_like(imgs)
        per_pixel = self.rng == "numpy"
        groups = {k: [] for k in WEATHER_TYPES}
        for b, c in enumerate(conditions):
            c = str(c)
            if c not in groups:
                raise ValueError(f"Unknown weather type: {c}")
            groups[c].append(b)
        # the reference handles samples one after another, so draws happen in sample order
        draws, seeds = {}, {}
        for b, c in enumerate(conditions):
            c = str(c)
            if c == "clean":
                continue
            if frame_ids is None:
                rs, seeds[b] = np.random, None
            else:
                rs, seeds[b] = self.frame_stream(frame_ids[b])
            if c == "fog":
                draws[b] = draw_fog(H, W, intensities[b], per_pixel, rs)
            elif c == "night":
                draws[b] = draw_night(H, W, intensities[b], per_pixel, rs)
            elif c == "rain":
                draws[b] = draw_rain(H, W, intensities[b], rs)
            elif c == "snow":
                draws[b] = draw_snow(H, W, intensities[b], rs)

        def philox(idx):
            return [seeds[b] if seeds[b] is not None else self._next_seed() for b in idx]
        dev = imgs.device
        if (not per_pixel and norm_out is not None and ops.weather_batch_ok(H, W) and B <= ops.WEATHER_BATCH_MAX and (H * W) % 4 == 0
                and (out is None or out.data_ptr() != imgs.data_ptr())):
            # throughput mode: every frame of the batch in ONE launch (awseg_weather_batch: a job per frame, the per-kind kernels' bodies,
            # identical bytes); snow frames that drew the 7x7 blur keep their own launch.  Seeds are drawn in the per-kind order (fog, night).
            wj = np.zeros(B, dtype=ops.N.WEATHER_JOB)
            n = 0
            rain_lists, snow_lists, snow7 = [], [], []
            roff = soff = 0
            fseeds, nseeds = philox(groups["fog"]), philox(groups["night"])
            for b in groups["clean"]:
                wj[n]["kind"], wj[n]["image"] = ops.N.WEATHER_CLEAN, b; n += 1
            for b, sd in zip(groups["fog"], fseeds):
                fj = ops.fog_jobs([b], [draws[b][1]], [sd])[0]
                wj[n]["kind"], wj[n]["image"], wj[n]["a"], wj[n]["b"], wj[n]["seed"] = ops.N.WEATHER_FOG, b, fj["beta"], fj["atmos"], fj["seed"]; n += 1
            for b, sd in zip(groups["night"], nseeds):
                nj = ops.night_jobs([b], [draws[b][1]], [draws[b][0]], [sd])[0]
                wj[n]["kind"], wj[n]["image"], wj[n]["a"], wj[n]["b"], wj[n]["seed"] = ops.N.WEATHER_NIGHT, b, nj["brightness"], nj["intensity"], nj["seed"]; n += 1
            for b in groups["rain"]:
                pl = np.asarray(draws[b][1], dtype=np.int32).reshape(-1, 5)
                wj[n]["kind"], wj[n]["image"], wj[n]["a"], wj[n]["prim_offset"], wj[n]["prim_count"] = ops.N.WEATHER_RAIN, b, float(draws[b][0]), roff, len(pl); n += 1
                rain_lists.append(pl); roff += len(pl)
            for b in groups["snow"]:
                if int(draws[b][2]) == 7:
                    snow7.append(b); continue
                pl = np.asarray(draws[b][1], dtype=np.int32).reshape(-1, 3)
                wj[n]["kind"], wj[n]["image"], wj[n]["a"], wj[n]["prim_offset"], wj[n]["prim_count"] = ops.N.WEATHER_SNOW, b, float(draws[b][0]), soff, len(pl); n += 1
                snow_lists.append(pl); soff += len(pl)
            rd = np.concatenate(rain_lists + [np.zeros((1, 5), np.int32)]) if rain_lists else None
            sf = np.concatenate(snow_lists + [np.zeros((1, 3), np.int32)]) if snow_lists else None
            if n == 0 or ops.weather_batch(imgs, wj[:n], rd, sf, norm_out, out=out):
                if out is not None and groups["clean"]:
                    out[groups["clean"]] = imgs[groups["clean"]]
                if snow7:
                    jobs, prims = ops.prim_jobs(snow7, [draws[b][0] for b in snow7], [draws[b][1] for b in snow7], [7] * len(snow7))
                    ops.snow(imgs, jobs, prims, out=out, norm_out=norm_out)
                return out
            seeds = {**seeds, **dict(zip(groups["fog"], fseeds)), **dict(zip(groups["night"], nseeds))}     # declined: per-kind calls, same seeds
        if groups["clean"]:
            idx = groups["clean"]
            if out is not None and out.data_ptr() != imgs.data_ptr():
                out[idx] = imgs[idx]
            if norm_out is not None:
                ops.normalize(imgs, out=norm_out, sel=torch.tensor(idx, dtype=torch.int32).to(dev, non_blocking=True))
        if groups["fog"]:
            idx = groups["fog"]
            jobs = ops.fog_jobs(idx, [draws[b][1] for b in idx], philox(idx))
            noise = torch.from_numpy(np.stack([draws[b][0] for b in idx])).to(dev, non_blocking=True) if per_pixel else None
            ops.fog(imgs, jobs, noise=noise, out=out, norm_out=norm_out)
        if groups["night"]:
            idx = groups["night"]
            jobs = ops.night_jobs(idx, [draws[b][1] for b in idx], [draws[b][0] for b in idx], philox(idx))
            noise = torch.from_numpy(np.stack([draws[b][2] for b in idx])).to(dev, non_blocking=True) if per_pixel else None
            ops.night(imgs, jobs, noise=noise, out=out, norm_out=norm_out)
        if groups["rain"] or groups["snow"]:
            tmp = out
            if out is not None and out.data_ptr() == imgs.data_ptr():
                tmp = torch.empty_like(imgs)                                         # blur reads neighbours: not in place
            if groups["rain"]:
                idx = groups["rain"]
                jobs, prims = ops.prim_jobs(idx, [draws[b][0] for b in idx], [draws[b][1] for b in idx])
                ops.rain(imgs, jobs, prims, out=tmp, norm_out=norm_out)
            if groups["snow"]:
                idx = groups["snow"]
                jobs, prims = ops.prim_jobs(idx, [draws[b][0] for b in idx], [draws[b][1] for b in idx], [draws[b][2] for b in idx])
                ops.snow(imgs, jobs, prims, out=tmp, norm_out=norm_out)
            if tmp is not out and out is not None:
                sel = groups["rain"] + groups["snow"]
                out[sel] = tmp[sel]
        return out

    # ---- pieces the reference exposes ----------------------------------------------------------
    def _generate_synthetic_depth(self, height: int, width: int) -> np.ndarray:
        """preprocessing.py:227-248 -> float64 [H,W] (numpy, like the reference)."""
        per_pixel = self.rng == "numpy"
        noise = torch.from_numpy(np.random.normal(0, 10, (height, width))[None]).to(self.device) if per_pixel else None
        jobs = ops.fog_jobs([0], [0.5], [self._next_seed()])
        return ops.synthetic_depth(height, width, jobs, self.device, noise)[0].cpu().numpy()

    def get_fog_density_map(self, image, depth=None):
        """preprocessing.py:250-288: (1 - contrast/p95(contrast)) * (0.3 + 0.7*depth/max(depth)),
        clipped to [0,1].  `image` float in [0,1] (the reference quantises it with (image*255)
        .astype(uint8), :270) or uint8; numpy in -> numpy out, device tensor in -> device tensor out."""
        is_np = isinstance(image, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(image)).to(self.device) if is_np else image
        if t.dtype != torch.uint8:
            t = (t * 255).to(torch.uint8)                                              # truncating cast, as numpy's astype
        h, w = t.shape[:2]
        if depth is None:
            depth = self._generate_synthetic_depth(h, w)
        d = torch.as_tensor(depth).to(self.device)
        contrast = ops.local_contrast(t.unsqueeze(0))[0]
        p95 = torch.quantile(contrast.flatten().double(), 0.95)                        # np.percentile's linear interpolation
        density = 1.0 - contrast.double() / (p95 + 1e-8)
        density = density * (0.3 + 0.7 * (d.double() / d.max().double()))
        out = density.clamp(0, 1)
        return out.cpu().numpy() if is_np else out


    def _apply_fog(self, image, intensity=None):
        return self.apply_weather_effect(self._as_u8(image), "fog", intensity)

    def _apply_rain(self, image, intensity=None):
        return self.apply_weather_effect(self._as_u8(image), "rain", intensity)

    def _apply_snow(self, image, intensity=None):
        return self.apply_weather_effect(self._as_u8(image), "snow", intensity)

    def _apply_night(self, image, intensity=None):
        return self.apply_weather_effect(self._as_u8(image), "night", intensity)

    @staticmethod
    def _as_u8(image):
        """The reference's private helpers take float32 images in [0,1] produced from uint8 by /255
        (:81); map them back exactly (x*255 is within 1e-5 of the original integer)."""
        if isinstance(image, np.ndarray) and image.dtype != np.uint8:
            return np.rint(image.astype(np.float64) * 255.0).astype(np.uint8)
        return image


class DepthEstimationPreprocessor:
    """PKG/data/preprocessing.py:291-411.  `estimate_depth` keeps the reference convention (one
    uint8 HWC numpy frame -> float64 [H,W] numpy); `estimate_depth_batch` is the device form the
    loader uses ([B,H,W,3] uint8 on the GPU -> float32 [B,H,W], no host round trip)."""

    def __init__(self, device: Union[str, torch.device] = "cuda") -> None:
        self.depth_model = None                                                      # :302
        self.device = torch.device(device)

    def estimate_depth(self, image):
        is_np = isinstance(image, np.ndarray)
        dev = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(self.device) if is_np else image
        out = ops.depth_estimate(dev.unsqueeze(0), dtype=torch.float64)[0]
        return out.cpu().numpy() if is_np else out

    _geometric_depth_estimation = estimate_depth                                      # :325

    def estimate_depth_batch(self, imgs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.depth_estimate(imgs, out=out, dtype=torch.float32)

    def depth_to_disparity(self, depth, baseline: float = 0.54):
        """:369-385."""
        if isinstance(depth, torch.Tensor):
            return baseline / torch.clamp(depth, min=1e-6)
        return baseline / np.maximum(depth, 1e-6)

    def preprocess_depth_for_training(self, depth, target_size) -> torch.Tensor:
        """:387-411: min-max normalise to [0,1] -> float32 tensor.  The reference resizes with
        cv2.resize (bilinear) when the shape differs; here that is torch's bilinear interpolation
        (align_corners=False, the same half-pixel sampling)."""
        t = torch.as_tensor(depth)
        if tuple(t.shape) != tuple(target_size):
            t = torch.nn.functional.interpolate(t[None, None].to(torch.float64), size=tuple(target_size), mode="bilinear",
                                                align_corners=False)[0, 0]
        t = (t - t.min()) / (t.max() - t.min() + 1e-8)
        return t.float()
