#!/usr/bin/env python3
"""Runs the general attention kernel (pt_attn_f16) at one shape, timed with hipEvents - default: the VAE mid block's single
512-wide head over one decode chunk at 576 x 1024 (8 frames x 9216 tokens); also for rocprofv3 --pmc runs.
    python tools/attn_general_one.py [nbatch S heads head_dim]      e.g. 2 257 16 80 = the CLIP ViT-H tower"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import ops
dev = torch.device("cuda:0")
nb, S, heads, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 9216, 1, 512)
C = heads * D
qkv = torch.randn(nb * S, 3 * C, device=dev, dtype=torch.float16)
run = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], nb, S, S, heads, D)
for _ in range(3):
    run()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 3)
print(f"pt_attn_f16 {nb} x {heads} heads x {S} tokens x head_dim {D}: {best:.3f} ms  {4.0 * nb * heads * S * S * D / best / 1e9:.0f} TFLOP/s")
