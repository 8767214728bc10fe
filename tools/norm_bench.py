#!/usr/bin/env python3
"""Times the GroupNorm / LayerNorm kernels at the 14x576x1024 level-0..2 shapes against their HBM byte counts (MI355X)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip
if os.environ.get("PT_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import ops

dev = torch.device("cuda:0")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (ns, rows, C) in [(28, 9216, 320), (28, 2304, 640), (28, 576, 1280), (28, 144, 1280), (2, 129024, 320), (2, 32256, 640), (2, 8064, 1280), (2, 2016, 1280)]:
    x = torch.randn(ns * rows, C, device=dev, dtype=torch.float16)
    g = torch.randn(C, device=dev, dtype=torch.float16); b = torch.randn(C, device=dev, dtype=torch.float16)
    us = timed(lambda: ops.groupnorm(x, g, b, rows_per_sample=rows, n_samples=ns, eps=1e-5, silu=True))
    nbytes = x.numel() * 2
    print(f"groupnorm+silu  samples={ns:3d} rows={rows:7d} C={C:5d}: {us:8.1f} us  {3 * nbytes / us / 1e6:6.2f} TB/s (2 reads + 1 write)")
    us = timed(lambda: ops.layernorm(x, g, b, 1e-5))
    print(f"layernorm       rows={ns * rows:7d} C={C:5d}: {us:8.1f} us  {2 * nbytes / us / 1e6:6.2f} TB/s (1 read + 1 write)")

for (B, F, S, heads) in [(2, 14, 9216, 5), (2, 14, 2304, 10), (2, 14, 576, 20)]:
    C = heads * 64
    qkv = torch.randn(B * F * S, 3 * C, device=dev, dtype=torch.float16)
    us = timed(lambda: ops.attn_temporal(qkv, B, F, S, heads, 64))
    nbytes = qkv.numel() * 2 + B * F * S * C * 2
    print(f"attn_temporal   B={B} F={F} S={S:5d} heads={heads:3d}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s (read qkv + write out)")
for n in [258048 * 320, 64512 * 640]:
    a = torch.randn(n, device=dev, dtype=torch.float16); r = torch.randn(n, device=dev, dtype=torch.float16)
    us = timed(lambda: ops.axpy(a, r, 2.0))
    print(f"axpy            n={n:10d}: {us:8.1f} us  {3 * n * 2 / us / 1e6:6.2f} TB/s (2 reads + 1 write)")
