"""Counter-based deterministic synthetic data (splitmix64 -> U[0,1) -> Box-Muller).

Both the build container and the GPU box regenerate identical inputs and weights
from (seed, element index) alone, independent of any torch RNG version
(SURVEY.md section 8d).  Used by bench.py, the golden-fixture maker and the tests.
"""
from __future__ import annotations

import math
from typing import Iterable, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_CHUNK = 1 << 22


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def name_seed(name: str, seed: int = 0) -> int:
    """FNV-1a 64-bit hash of a tensor name, mixed with a base seed."""
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def _u01(seed: int, start: int, n: int, stream: int) -> np.ndarray:
    """n doubles in [0,1) for counters start..start+n-1 of (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
                           + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))[0]
        ctr = np.arange(start, start + n, dtype=np.uint64) + base
    bits = _splitmix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform_np(n: int, seed: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        m = min(_CHUNK, n - s)
        out[s:s + m] = (lo + (hi - lo) * _u01(seed, s, m, 0)).astype(np.float32)
    return out


def normal_np(n: int, seed: int, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        m = min(_CHUNK, n - s)
        u1 = 1.0 - _u01(seed, s, m, 1)          # (0,1]
        u2 = _u01(seed, s, m, 2)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
        out[s:s + m] = (mean + std * z).astype(np.float32)
    return out


def uniform(shape: Iterable[int], seed: int, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    shape = tuple(shape)
    return torch.from_numpy(uniform_np(int(np.prod(shape)), seed, lo, hi)).view(shape)


def normal(shape: Iterable[int], seed: int, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    shape = tuple(shape)
    return torch.from_numpy(normal_np(int(np.prod(shape)), seed, mean, std)).view(shape)


def randint(shape: Iterable[int], seed: int, lo: int, hi: int) -> torch.Tensor:
    """Integers in [lo, hi)."""
    shape = tuple(shape)
    u = _u01(seed, 0, int(np.prod(shape)), 3)
    return torch.from_numpy((lo + np.floor(u * (hi - lo))).astype(np.int64)).view(shape)


# per-key gain overrides: raw sensor units enter lidar/radar conv1 (intensity up to 255), residual
# sums grow per block, and the head's last 1x1 should land logits in the sigmoid's active range.
_GAINS = (("lidar_encoder.conv1.weight", 0.03), ("_head.2.weight", 1.0), ("lidar_init.2.weight", 1.0),
          ("fusion_fc.weight", 1.0))


@torch.no_grad()
def fill_state_dict_(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter/buffer of `module` with name-keyed synthetic values.

    Scales are chosen so activations stay O(1..30) through the ~25 layers of the path
    (relative-error comparisons stay meaningful, the heatmap is not saturated) and BN
    running statistics are non-trivial.  Depends only on (key name, shape, seed).
    """
    sd = module.state_dict()
    for key, t in sd.items():
        s = name_seed(key, seed)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t.fill_(3)
        elif leaf == "running_mean":
            t.copy_(normal(t.shape, s, 0.0, 0.1))
        elif leaf == "running_var":
            t.copy_(uniform(t.shape, s, 0.5, 1.5))
        elif t.dim() == 1 and leaf == "weight":       # norm scale (last BN of a residual block: small)
            lo, hi = (0.2, 0.5) if key.endswith("bn2.weight") and "layer" in key else (0.6, 1.2)
            t.copy_(uniform(t.shape, s, lo, hi))
        elif leaf == "bias":
            t.copy_(normal(t.shape, s, 0.0, 0.05))
        else:
            fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
            gain = math.sqrt(2.0)
            for pat, g in _GAINS:
                if key.endswith(pat):
                    gain = g
            t.copy_(normal(t.shape, s, 0.0, gain * math.sqrt(1.0 / max(fan_in, 1))))


def frame_inputs(batch: int, n_cams: int, height: int, width: int, n_points: int,
                 point_channels: int = 4, n_radars: int = 0, radar_points: int = 125,
                 radar_channels: int = 7, seed: int = 0x5EED
                 ) -> Tuple[torch.Tensor, torch.Tensor, list]:
    """Synthetic frame per SURVEY.md 8d: images N(0,1); LiDAR x,y~U(+-51.2), z~U(-5,3),
    intensity~U(0,255) (extra channels U(0,1)); radar N(0,1)."""
    imgs = normal((batch, n_cams, 3, height, width), seed) if n_cams else None
    pts = None
    if n_points:
        cols = [uniform((batch, n_points), seed + 11, -51.2, 51.2),
                uniform((batch, n_points), seed + 12, -51.2, 51.2),
                uniform((batch, n_points), seed + 13, -5.0, 3.0),
                uniform((batch, n_points), seed + 14, 0.0, 255.0)]
        for c in range(4, point_channels):
            cols.append(uniform((batch, n_points), seed + 11 + c, 0.0, 1.0))
        pts = torch.stack(cols[:point_channels], dim=2).contiguous()
    radars = [normal((batch, radar_points, radar_channels), seed + 101 + r) for r in range(n_radars)]
    return imgs, pts, radars


def gt_boxes(batch: int, n_boxes: int, seed: int = 0x5EED) -> Tuple[torch.Tensor, torch.Tensor]:
    """GT per SURVEY.md 8d config 4: centres U(range), sizes U(0.5,5), yaw U(-pi,pi), labels U{0..9}."""
    b = torch.empty(batch, n_boxes, 9)
    b[..., 0] = uniform((batch, n_boxes), seed + 201, -51.2, 51.2)
    b[..., 1] = uniform((batch, n_boxes), seed + 202, -51.2, 51.2)
    b[..., 2] = uniform((batch, n_boxes), seed + 203, -5.0, 3.0)
    b[..., 3:6] = uniform((batch, n_boxes, 3), seed + 204, 0.5, 5.0)
    b[..., 6] = uniform((batch, n_boxes), seed + 205, -math.pi, math.pi)
    b[..., 7:9] = normal((batch, n_boxes, 2), seed + 206, 0.0, 2.0)
    labels = randint((batch, n_boxes), seed + 207, 0, 10)
    return b, labels
