"""Deterministic, portable (integer-hash) tensor fill used by BOTH the golden generator (applied to the
reference's modules, authoring container only) and the tests (applied to pixelpick_amd's modules), so
that 23 MB of weights never have to be stored: the fixtures hold outputs, losses and gradient summaries.
"""
import zlib

import numpy as np
import torch


def _uniform01(n: int, seed: int) -> np.ndarray:
    """n floats in [0,1): SplitMix64 on the element index (exact integer arithmetic)."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24))


def fill(shape, key: str, lo: float, hi: float) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _uniform01(n, zlib.crc32(key.encode()))
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(tuple(shape))


def formula_state_dict(template: dict, salt: str = "") -> dict:
    """template: a state_dict (reference layout: conv OIHW) -> same keys/shapes with formula values.  `salt` selects
    another member of the family (the well-conditioned fixtures search over it, tools/gen_golden_net_tight.py)."""
    out = {}
    for k0, v in template.items():
        k = k0 + salt
        shape = tuple(v.shape)
        if k0.endswith("num_batches_tracked"):
            out[k0] = torch.zeros((), dtype=torch.long)
        elif k0.endswith("running_mean"):
            out[k0] = fill(shape, k, -0.2, 0.2)
        elif k0.endswith("running_var"):
            out[k0] = fill(shape, k, 0.6, 1.6)
        elif v.dim() == 4:                               # conv weight OIHW: variance-preserving uniform
            fan_in = shape[1] * shape[2] * shape[3]
            a = float(np.sqrt(6.0 / fan_in))             # He-uniform
            out[k0] = fill(shape, k, -a, a)
        elif k0.endswith("weight"):                      # BN / GN gamma
            out[k0] = fill(shape, k, 0.7, 1.3)
        else:                                            # biases, BN beta
            out[k0] = fill(shape, k, -0.2, 0.2)
    return out


def formula_input(B, H, W, key="x") -> torch.Tensor:
    return fill((B, 3, H, W), key, -1.5, 1.5)


def formula_labels(B, H, W, n_classes, ignore_index, n_per_image, key="y") -> torch.Tensor:
    y = torch.full((B, H, W), ignore_index, dtype=torch.int64)
    for b in range(B):
        u = _uniform01(2 * n_per_image, zlib.crc32(f"{key}{b}".encode()))
        pos = np.unique((u[:n_per_image] * H * W).astype(np.int64))
        lab = (u[n_per_image:n_per_image + len(pos)] * n_classes).astype(np.int64)
        y[b].view(-1)[torch.from_numpy(pos)] = torch.from_numpy(lab)
    return y


def summarize(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()], dtype=np.float64)


def tie_aliases(sd: dict) -> dict:
    """MobileNetV2 registers its layers three times (`features`, `low_level_features` = features[0:4], `high_level_features` =
    features[4:], mobilenet_v2.py:118-126), so a real state_dict carries every backbone tensor under two names with ONE value:
    make a per-key formula dict consistent the same way (the `features.N` entry wins)."""
    for k in list(sd.keys()):
        for alias in (".low_level_features.", ".high_level_features."):
            if alias in k:
                sd[k] = sd[k.replace(alias, ".features.")].clone()
    return sd
