#!/usr/bin/env python3
"""Runs the spatial attention kernel at the 14x576x1024 level-0 shape (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import ops
dev = torch.device("cuda:0")
Nimg, S, heads = 28, 9216, 5
qkv = torch.randn(Nimg * S, 3 * heads * 64, device=dev, dtype=torch.float16)
for _ in range(4):
    o = ops.attn_spatial(qkv, Nimg, S, heads, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    o = ops.attn_spatial(qkv, Nimg, S, heads, 64)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"attn_spatial {Nimg}x{heads}x{S}x64: {ms:.3f} ms  {4.0 * Nimg * heads * S * S * 64 / ms / 1e9:.0f} TFLOP/s")
