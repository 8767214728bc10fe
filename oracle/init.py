"""Deterministic, name-keyed parameter initialisation shared by the golden generator and the tests.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Weights are never stored in fixtures: both sides rebuild them
from (seed, parameter name, shape), so a fixture only holds inputs and expected outputs.
"""
from __future__ import annotations

import math
import zlib

import torch


def seeded_tensor(name: str, shape, seed: int, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std + mean


@torch.no_grad()
def seeded_init_(module: torch.nn.Module, seed: int, zero_conv_std: float = 0.05) -> torch.nn.Module:
    """Every parameter <- N(mean, std) keyed by its state-dict name.  Zero-initialised ControlNet output
    convs are randomised too (otherwise the whole ControlNet branch is a no-op)."""
    for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "mix_factor":
            t = seeded_tensor(name, p.shape, seed, std=0.7)
        elif "norm" in name and leaf == "weight":
            t = seeded_tensor(name, p.shape, seed, std=0.1, mean=1.0)
        elif leaf == "bias":
            t = seeded_tensor(name, p.shape, seed, std=0.05)
        else:
            fan_in = p[0].numel() if p.ndim > 1 else p.numel()
            std = 1.0 / math.sqrt(fan_in)
            if name.startswith(("controlnet_down_blocks", "controlnet_mid_block")) or name.endswith("controlnet_cond_embedding.conv_out.weight"):
                std = max(std, zero_conv_std)
            t = seeded_tensor(name, p.shape, seed, std=std)
        p.copy_(t.to(p.dtype))
    return module
