#!/usr/bin/env python3
"""Runs each tile configuration of the implicit-GEMM kernel a few times on one shape (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_linear
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16128, 3840, 1280)))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).half().to(dev)
w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
pw = pack_linear(w, None, dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
for cfg in (0, 1, 2, 3, 4):
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    for _ in range(3):
        ops.igemm(x, pw, out=out)
torch.cuda.synchronize()
