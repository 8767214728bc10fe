"""Oracle restatement of ``transformers.CLIPVisionModelWithProjection`` - the ``image_encoder`` of the reference pipeline
(``/root/reference/pipeline/pipeline_stable_video_diffusion_controlnet.py:22,125``; called at ``:157``:
``self.image_encoder(image).image_embeds`` on the 224 x 224 anti-aliased resize of the conditioning image, NOT normalised
with the CLIP mean / std - SURVEY Q7).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Pinning status: PINNED.  ``transformers`` (``/root/reference/requirements.txt:14``) is installed in the build image, so
``tests/golden/make_golden.py: gen_clip`` runs the very class the reference imports (random-init ``CLIPVisionConfig``s: a
small one with ViT-H's head_dim 80 / patch 14 / 257 tokens for both activations, and ViT-H/14's real widths - hidden 1280,
16 heads, MLP 5120, projection 1024 - at 2 layers) and ``tests/test_oracle_golden.py`` holds this restatement to those outputs
(``tests/golden/clip.npz``).  Parameter names equal the transformers state-dict keys, so both sides rebuild the same weights
from ``(seed, name, shape)``.

Algorithm (transformers ``models/clip/modeling_clip.py``): patch embedding = Conv2d(3, C, kernel = stride = patch, no bias)
-> ``[B, 256, C]``; class token prepended; + learned position embedding (257 rows); ``pre_layrnorm`` (sic); per layer
``x += out_proj(softmax(q k^T / sqrt(d)) v)`` on ``layer_norm1(x)`` with biased q / k / v / out projections, then
``x += fc2(act(fc1(layer_norm2(x))))``; ``post_layernorm`` of the CLASS token; ``visual_projection`` (no bias).
``act``: "gelu" (erf; the laion ViT-H/14 SVD ships) or "quick_gelu" (``x * sigmoid(1.702 x)``; the OpenAI checkpoints).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .quant import q


def vit_h_config(**over):
    """``image_encoder/config.json`` of stabilityai/stable-video-diffusion-img2vid = laion/CLIP-ViT-H-14-laion2B-s32B-b79K's vision
    tower [UNVERIFIED-MEMORY: the file is not in the reference tree]."""
    cfg = dict(hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=32, num_attention_heads=16,
               num_channels=3, image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-5)
    cfg.update(over)
    return cfg


def tiny_clip_config(**over):
    """ViT-H's head_dim (80), patch (14) and token count (257) at a width the CPU suite runs in a second."""
    cfg = dict(hidden_size=160, intermediate_size=640, projection_dim=64, num_hidden_layers=3, num_attention_heads=2,
               num_channels=3, image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-5)
    cfg.update(over)
    return cfg


def act_fn(name):
    if name == "gelu":
        return F.gelu
    if name == "quick_gelu":
        return lambda x: x * torch.sigmoid(1.702 * x)
    raise ValueError(f"hidden_act {name!r}")


class CLIPAttention(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        b, s, c = x.shape
        d = c // self.heads
        sp = lambda t: q(t, True).view(b, s, self.heads, d).transpose(1, 2)
        qq, k, v = sp(self.q_proj(x)), sp(self.k_proj(x)), sp(self.v_proj(x))
        w = q(torch.softmax(q(qq @ k.transpose(-1, -2) / math.sqrt(d)), dim=-1))
        o = q((w @ v).transpose(1, 2).reshape(b, s, c), True)
        return self.out_proj(o)


class CLIPMLP(nn.Module):
    def __init__(self, c, inter, act):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, inter), nn.Linear(inter, c)
        self.act = act_fn(act)

    def forward(self, x):
        return self.fc2(q(self.act(q(self.fc1(x), True)), True))


class CLIPEncoderLayer(nn.Module):
    def __init__(self, c, heads, inter, act, eps):
        super().__init__()
        self.self_attn = CLIPAttention(c, heads)
        self.layer_norm1 = nn.LayerNorm(c, eps=eps)
        self.mlp = CLIPMLP(c, inter, act)
        self.layer_norm2 = nn.LayerNorm(c, eps=eps)

    def forward(self, x):
        # the residual stream is an fp16 pair on the MI355X path (kind "rb": oracle/quant.py); norms read its high half
        x = q(x + self.self_attn(q(self.layer_norm1(q(x, True)), True)), True, wide="rb")
        return q(x + self.mlp(q(self.layer_norm2(q(x, True)), True)), True, wide="rb")


class CLIPVisionEmbeddings(nn.Module):
    def __init__(self, c, channels, image_size, patch):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(c))
        self.patch_embedding = nn.Conv2d(channels, c, kernel_size=patch, stride=patch, bias=False)
        n = (image_size // patch) ** 2 + 1
        self.position_embedding = nn.Embedding(n, c)
        self.image_size = image_size

    def forward(self, pixel_values):
        b, _, h, w = pixel_values.shape
        if h != self.image_size or w != self.image_size:
            raise ValueError(f"Input image size ({h}*{w}) doesn't match model ({self.image_size}*{self.image_size}).")
        p = self.patch_embedding(q(pixel_values, True)).flatten(2).transpose(1, 2)
        e = torch.cat([self.class_embedding.expand(b, 1, -1), p], dim=1)
        return q(e + self.position_embedding.weight[None], True)


class CLIPVisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.hidden_size
        self.embeddings = CLIPVisionEmbeddings(c, cfg.num_channels, cfg.image_size, cfg.patch_size)
        self.pre_layrnorm = nn.LayerNorm(c, eps=cfg.layer_norm_eps)
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([CLIPEncoderLayer(c, cfg.num_attention_heads, cfg.intermediate_size, cfg.hidden_act,
                                                              cfg.layer_norm_eps) for _ in range(cfg.num_hidden_layers)])
        self.post_layernorm = nn.LayerNorm(c, eps=cfg.layer_norm_eps)

    def forward(self, pixel_values):
        x = q(self.pre_layrnorm(self.embeddings(pixel_values)), True)
        for layer in self.encoder.layers:
            x = layer(x)
        return x, q(self.post_layernorm(q(x[:, 0], True)), True)


class CLIPVisionModelWithProjection(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        self.config = SimpleNamespace(**cfg)
        self.vision_model = CLIPVisionTransformer(self.config)
        self.visual_projection = nn.Linear(cfg["hidden_size"], cfg["projection_dim"], bias=False)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def forward(self, pixel_values):
        last, pooled = self.vision_model(pixel_values)
        return SimpleNamespace(image_embeds=q(self.visual_projection(pooled), True), last_hidden_state=last)
