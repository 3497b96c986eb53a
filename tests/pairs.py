"""Deterministic synthetic frame pairs shared by the golden generator, the tests,
``__graft_entry__.smoke()`` and ``bench.py`` (SURVEY.md §8d "Synthetic inputs")."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def random_pair(b: int, h: int, w: int, seed: int):
    """i.i.d. U[0,1) frames: worst case for warp locality."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, 3, h, w, generator=g), torch.rand(b, 3, h, w, generator=g)


def smooth_pair(b: int, h: int, w: int, seed: int):
    """Low-pass noise + a little grain; frame 1 is frame 0 shifted by (-3, +4) px, so
    flows, masks and attention are non-degenerate."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(b, 3, h // 8 + 4, w // 8 + 4, generator=g)
    big = F.interpolate(base, size=(h + 32, w + 32), mode="bicubic", align_corners=True).clamp(0, 1)
    big = (big + 0.15 * torch.rand(b, 3, h + 32, w + 32, generator=g)).clamp(0, 1)
    return (big[:, :, 16:16 + h, 16:16 + w].contiguous(),
            big[:, :, 13:13 + h, 20:20 + w].contiguous())


def mixed_batch(b: int, h: int, w: int, seed: int):
    """A batch whose samples differ in kind: i.i.d. frames (seed), i.i.d. frames (seed + 1), smooth shifted frames (seed), ...
    Used for the ensemble fixture whose samples pick different pyramid levels (network_base.py:593-603)."""
    parts = []
    for i in range(b):
        parts.append(random_pair(1, h, w, seed + i) if i % 3 < 2 else smooth_pair(1, h, w, seed + i - 2))
    return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)


PAIR_KINDS = {"smooth": smooth_pair, "random": random_pair, "mixed": mixed_batch}


def uint8_pair(h: int, w: int, seed: int = 0):
    """Demo-path input: HWC uint8 frames."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    t = torch.from_numpy(a).permute(2, 0, 1)[None]
    big = F.interpolate(t, size=(h + 8, w + 8), mode="bilinear", align_corners=True)[0].permute(1, 2, 0).numpy()
    noise = rng.integers(-12, 13, big.shape)
    big = np.clip(np.round(big + noise), 0, 255).astype(np.uint8)
    return big[4:4 + h, 4:4 + w].copy(), big[2:2 + h, 7:7 + w].copy()


def uint8_video(n: int, h: int, w: int, seed: int = 0):
    """n consecutive uint8 [H,W,3] frames of a short synthetic clip: a smooth texture drifting by (2, -3) px per frame with fresh
    grain on every frame (the video-loop tests: demo_2x.py:129-168)."""
    rng = np.random.default_rng(seed)
    m = 4 * n + 8
    a = rng.integers(0, 256, (h // 8 + m // 8 + 3, w // 8 + m // 8 + 3, 3)).astype(np.float32)
    t = torch.from_numpy(a).permute(2, 0, 1)[None]
    big = F.interpolate(t, size=(h + m, w + m), mode="bilinear", align_corners=True)[0].permute(1, 2, 0).numpy()
    out = []
    for i in range(n):
        y, x = 4 + 2 * i, m - 4 - 3 * i
        crop = big[y:y + h, x:x + w]
        out.append(np.clip(np.round(crop + rng.integers(-10, 11, crop.shape)), 0, 255).astype(np.uint8))
    return out
