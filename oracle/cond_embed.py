"""Oracle restatement of the trajectory / camera condition encoders.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PINNED against the reference classes
(``tests/golden/cond_embed_*.npz``).

* ``models/controlnet_sdv.py:61-116``            ControlNetConditioningEmbeddingSVD
* ``models/controlnet_sdv_cam_infer.py:61-130``  ControlNetConditioningEmbeddingSVD_CAM
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .quant import q


def zero_module(m: nn.Module) -> nn.Module:
    """``controlnet_sdv.py:860-863``."""
    for p in m.parameters():
        nn.init.zeros_(p)
    return m


class ControlNetConditioningEmbeddingSVD(nn.Module):
    """conv3x3(3->c0) SiLU ; for each level: conv3x3(c->c) SiLU, conv3x3 s2 (c->c') SiLU ; zero-init conv3x3 -> C.
    Input ``[B, F, 3, H, W]`` is viewed as ``[B*F, 3, H, W]``; output ``[B*F, C, H/8, W/8]``."""

    def __init__(self, conditioning_embedding_channels, conditioning_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        ch = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(conditioning_channels, ch[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for a, b in zip(ch[:-1], ch[1:]):
            self.blocks.append(nn.Conv2d(a, a, 3, padding=1))
            self.blocks.append(nn.Conv2d(a, b, 3, padding=1, stride=2))
        self.conv_out = zero_module(nn.Conv2d(ch[-1], conditioning_embedding_channels, 3, padding=1))

    def features(self, conditioning):
        b, f, c, h, w = conditioning.shape
        e = q(F.silu(q(self.conv_in(q(conditioning.reshape(b * f, c, h, w), True)))), True)
        for blk in self.blocks:
            e = q(F.silu(q(blk(e))), True)
        return e

    def forward(self, conditioning):
        return q(self.conv_out(self.features(conditioning)), True)


class ControlNetConditioningEmbeddingSVD_CAM(ControlNetConditioningEmbeddingSVD):
    """Camera twin (``controlnet_sdv_cam_infer.py:84,96-130``): before ``conv_out`` the 12-vector R|T of each frame
    is tiled over the 1/8-res map, concatenated on channels and projected back per pixel by Linear(c+12 -> c)."""

    def __init__(self, conditioning_embedding_channels, conditioning_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__(conditioning_embedding_channels, conditioning_channels, block_out_channels)
        self.block_out_channels = block_out_channels
        self.cc_projection = nn.Linear(block_out_channels[-1] + 12, block_out_channels[-1])

    def forward(self, conditioning, camera_RT=None):
        b, f = conditioning.shape[:2]
        e = self.features(conditioning)
        if camera_RT is not None:
            cam = camera_RT.reshape(b * f, camera_RT.shape[-1])[:, :, None, None]
            cam = cam.repeat(1, 1, e.shape[2], e.shape[3])
            e = torch.cat((e, q(cam, True)), dim=1).permute(0, 2, 3, 1)
            e = q(self.cc_projection(e), True).permute(0, 3, 1, 2)
        return q(self.conv_out(e), True)
