#!/usr/bin/env python3
"""Runs the spatial attention kernel at the 14x576x1024 level-0 shape (for rocprofv3 --pmc runs).
PT_LIB=<path> loads an experimental build of libposetraj_hip.so instead of the in-tree one."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip
if os.environ.get("PT_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import ops
dev = torch.device("cuda:0")
Nimg, S, heads = 28, 9216, 5
if len(sys.argv) > 1:
    Nimg, S, heads = (int(v) for v in sys.argv[1:4])
if os.environ.get("ATTN_CHECK"):                   # correctness of an experimental build: ragged and aligned S, large-score rows, vs fp32 softmax
    for (n_, s_, h_, amp) in ((2, 1000, 5, 1.0), (1, 2304, 10, 1.0), (1, 577, 5, 6.0), (3, 64, 5, 1.0), (2, 1100, 5, 1.0), (1, 1297, 3, 6.0), (1, 1024, 2, 1.0), (1, 4100, 1, 3.0), (2, 40, 5, 1.0), (2, 129, 5, 2.0), (1, 200, 3, 8.0)):
        g = torch.Generator().manual_seed(s_)
        x = (torch.randn(n_ * s_, 3 * h_ * 64, generator=g) * amp).half().to(dev)
        got = ops.attn_spatial(x, n_, s_, h_, 64).float()
        q, k, v = (x.float().view(n_, s_, 3, h_, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ref = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v
        ref = ref.permute(0, 2, 1, 3).reshape(n_ * s_, h_ * 64)
        print(f"check N={n_} S={s_} heads={h_} amp={amp}: rel-L2 {float((got - ref).norm() / ref.norm()):.3e}  max |d| {float((got - ref).abs().max()):.3e}")
qkv = torch.randn(Nimg * S, 3 * heads * 64, device=dev, dtype=torch.float16)
if os.environ.get("ATTN_ZEROS"):                   # power check: zero operands (MI355X_MICROARCH.md, DVFS give-back)
    qkv.zero_()
for _ in range(6):
    o = ops.attn_spatial(qkv, Nimg, S, heads, 64, q_prescaled=bool(os.environ.get("ATTN_PRE")))
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        o = ops.attn_spatial(qkv, Nimg, S, heads, 64, q_prescaled=bool(os.environ.get("ATTN_PRE")))
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
print(f"{os.environ.get('PT_LIB', 'in-tree')} attn_spatial {Nimg}x{heads}x{S}x64: {best:.3f} ms  "
      f"{4.0 * Nimg * heads * S * S * 64 / best / 1e9:.0f} TFLOP/s")
