#!/usr/bin/env python3
"""Per-op error budget on the MI355X (diagnostic): every HIP entry point of the path on fp16-exact inputs against the
same op in fp64 PyTorch, split into  out = rel-L2(fp16(ref), ref)  - what ONE rounding of the exact result costs, the
floor of any fp16-storing kernel -  and  impl = rel-L2(hip, fp16(ref))  - what the kernel adds on top (internal fp16
operands such as the attention P matrix, accumulation order, transcendental approximations; ~0 for a kernel that is
exact up to its output rounding).  hip|ref = rel-L2(hip, ref) is what tests/test_kernels_gpu.py bounds."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import ops  # noqa: E402
from posetraj_amd.packing import pack_conv2d, pack_conv_t3, pack_linear, vec16  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def h16(*shape, scale=1.0, mean=0.0):
    return (torch.randn(*shape, generator=g) * scale + mean).half().to(dev)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def report(name, y, ref):
    r16 = ref.half().double() if ref.dtype != torch.float16 else ref.double()
    print(f"{name:58s} hip|ref {rel(y, ref):.2e}   out {rel(r16, ref):.2e}   impl {rel(y, r16):.2e}", flush=True)


D = lambda t: t.double().cpu()

# ---- linear layers at the level-0 shapes
for (M, N, K) in [(4096, 320, 320), (4096, 960, 320), (4096, 320, 1280), (2048, 1280, 5120)]:
    x, w, b = h16(M, K), h16(N, K, scale=K ** -0.5), h16(N)
    report(f"linear {M}x{N}x{K}", ops.igemm(x, pack_linear(w, b, dev)), F.linear(D(x), D(w), D(b)))
M, N, K = 4096, 2560, 320
x, w, b = h16(M, K), h16(N, K, scale=K ** -0.5), h16(N)
hh, gg = F.linear(D(x), D(w), D(b)).chunk(2, dim=-1)
report(f"GEGLU linear {M}x{N}x{K}", ops.igemm(x, pack_linear(w, b, dev, geglu=True)), hh * F.gelu(gg))
res, vec, blend = h16(M, 320), h16(4, 320), h16(M, 320)
x2, w2, b2 = h16(M, 1280), h16(320, 1280, scale=1280 ** -0.5), h16(320)
y = ops.igemm(x2, pack_linear(w2, b2, dev), res=res, vec=vec, vec_mode=1, vG=M // 4)
report("linear + res + row vector", y, F.linear(D(x2), D(w2), D(b2)) + D(res) + D(vec).repeat_interleave(M // 4, 0))
y = ops.igemm(x2, pack_linear(w2, b2, dev), res=res, blend=blend, alpha=0.3)
report("linear + res + AlphaBlender", y, 0.3 * D(blend) + 0.7 * (F.linear(D(x2), D(w2), D(b2)) + D(res)))
# ---- convolutions
Nn, H, W, Ci, Co = 4, 24, 32, 320, 320
x = h16(Nn, H, W, Ci)
w, b = h16(Co, Ci, 3, 3, scale=(9 * Ci) ** -0.5), h16(Co)
report("conv3x3 320->320", ops.igemm(x, pack_conv2d(w, b, dev), geom=(Nn, H, W)).view(Nn, H, W, Co),
       F.conv2d(D(x).permute(0, 3, 1, 2), D(w), D(b), padding=1).permute(0, 2, 3, 1))
wt = h16(Co, Ci, 3, 1, 1, scale=(3 * Ci) ** -0.5)
xt = x.view(1, Nn, H * W, Ci)
ref = F.conv3d(D(x).view(1, Nn, H, W, Ci).permute(0, 4, 1, 2, 3), D(wt), D(b), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1)
report("temporal conv (3,1,1) 320->320", ops.igemm(xt, pack_conv_t3(wt, b, dev), geom=(1, Nn, H * W)).view(1, Nn, H, W, Co), ref)
# ---- norms
for C, rows in [(320, 768), (1280, 144)]:
    x = h16(4 * rows, C, scale=1.5, mean=0.4)
    ga, be = h16(C, scale=0.1, mean=1.0), h16(C, scale=0.1)
    y = ops.groupnorm(x, ga, be, rows_per_sample=rows, n_samples=4, eps=1e-5, silu=True)
    ref = F.silu(F.group_norm(D(x).view(4, rows, C).permute(0, 2, 1), 32, D(ga), D(be), eps=1e-5)).permute(0, 2, 1).reshape(4 * rows, C)
    report(f"GroupNorm+SiLU C={C}", y, ref)
    y = ops.layernorm(x, ga, be, 1e-5)
    report(f"LayerNorm C={C}", y, F.layer_norm(D(x), (C,), D(ga), D(be), 1e-5))
# ---- attention
for (Nimg, S, heads) in [(2, 576, 5), (1, 2304, 5), (1, 9216, 1)]:
    C = heads * 64
    qkv = h16(Nimg * S, 3 * C)
    q, k, v = [D(t).view(Nimg, S, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Nimg * S, C)
    report(f"spatial attention S={S} heads={heads} (unit-variance q,k)", ops.attn_spatial(qkv, Nimg, S, heads, 64), ref)
    qkv2 = qkv.clone(); qkv2[:, :2 * C] *= 0.35                      # scores ~ N(0, 1): soft attention over many keys
    q, k, v = [D(t).view(Nimg, S, heads, 64).transpose(1, 2) for t in qkv2.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Nimg * S, C)
    report(f"spatial attention S={S} heads={heads} (soft)", ops.attn_spatial(qkv2, Nimg, S, heads, 64), ref)
B, Fr, S, heads = 2, 14, 128, 5
C = heads * 64
qkv = h16(B * Fr * S, 3 * C)
seq = lambda t: D(t).view(B, Fr, S, heads, 64).permute(0, 2, 3, 1, 4).reshape(B * S, heads, Fr, 64)
q, k, v = [seq(t) for t in qkv.chunk(3, dim=-1)]
r = F.scaled_dot_product_attention(q, k, v)
report("temporal attention F=14", ops.attn_temporal(qkv, B, Fr, S, heads, 64),
       r.view(B, S, heads, Fr, 64).permute(0, 3, 1, 2, 4).reshape(B * Fr * S, C))
