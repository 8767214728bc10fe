"""Weight pre-packing: reference (diffusers-format) state-dict tensors -> the fp16 ``[Npad, Kpad]`` images the HIP
kernels stream (``include/posetraj_hip.h`` conventions).  Pure layout work (permute / pad / concatenate), done once
at load time on whatever device the source tensors live on."""
from __future__ import annotations

from typing import Optional

import torch

from .ops import Packed

BN, BK = 128, 64


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _finish(w2d: torch.Tensor, bias: Optional[torch.Tensor], device, **kw) -> Packed:
    N, K = w2d.shape
    Np, Kp = _round_up(N, BN), _round_up(K, BK)
    w = torch.zeros((Np, Kp), dtype=torch.float16, device=device)
    w[:N, :K] = w2d.to(device=device, dtype=torch.float16)
    b = None
    if bias is not None:
        b = torch.zeros((Np,), dtype=torch.float16, device=device)
        b[:N] = bias.to(device=device, dtype=torch.float16)
    return Packed(w=w, bias=b, N=N, K=K, **kw)


def _geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """rows [value(0..I) | gate(0..I)] -> blocks of 16: value[0:16], gate[0:16], value[16:32], gate[16:32], ...
    so that a 32-column slab of the GEMM tile holds both operands of 16 outputs."""
    two_i = t.shape[0]
    inner = two_i // 2
    if inner % 16:
        raise ValueError(f"GEGLU inner width {inner} must be a multiple of 16")
    val = t[:inner].reshape(inner // 16, 16, *t.shape[1:])
    gate = t[inner:].reshape(inner // 16, 16, *t.shape[1:])
    return torch.stack([val, gate], dim=1).reshape(two_i, *t.shape[1:])


def pack_linear(w: torch.Tensor, bias: Optional[torch.Tensor], device, geglu: bool = False) -> Packed:
    w = w.detach().float()
    bias = None if bias is None else bias.detach().float()
    if geglu:
        w = _geglu_interleave(w)
        bias = None if bias is None else _geglu_interleave(bias)
    return _finish(w, bias, device, cin=w.shape[1], geglu=geglu)


def pack_conv2d(w: torch.Tensor, bias: Optional[torch.Tensor], device, stride: int = 1, padding: int = 1) -> Packed:
    """``[Co, Ci, kh, kw]`` -> ``[Co, kh*kw*Cip]`` with Ci zero-padded to a multiple of 8."""
    co, ci, kh, kw = w.shape
    cip = _round_up(ci, 8)
    wp = torch.zeros((co, kh, kw, cip), dtype=torch.float32, device=w.device)
    wp[..., :ci] = w.detach().float().permute(0, 2, 3, 1)
    return _finish(wp.reshape(co, kh * kw * cip), None if bias is None else bias.detach().float(), device,
                   KH=kh, KW=kw, stride=stride, pad_h=padding, pad_w=padding, cin=cip)


def pack_conv_t3(w: torch.Tensor, bias: Optional[torch.Tensor], device) -> Packed:
    """Conv3d ``[Co, Ci, 3, 1, 1]`` (padding (1,0,0)) -> a (3 x 1) convolution over the image (H', W') = (F, H*W)."""
    co, ci, kt, kh, kw = w.shape
    if (kt, kh, kw) != (3, 1, 1) or ci % 8:
        raise ValueError(f"unsupported temporal conv weight {tuple(w.shape)}")
    wp = w.detach().float().reshape(co, ci, 3).permute(0, 2, 1).reshape(co, 3 * ci)
    return _finish(wp, None if bias is None else bias.detach().float(), device, KH=3, KW=1, stride=1, pad_h=1, pad_w=0,
                   cin=ci)


def vec16(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float16).contiguous()
