"""pt_colsum_f16 (bias gradients) on the training step's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posetraj_amd import autodiff as AD
dev = torch.device("cuda:0")
for M, C in ((40320, 320), (40320, 2560), (10080, 640), (10080, 5120), (2520, 1280), (630, 1280), (2580480, 16)):
    dy = torch.randn(M, C, device=dev).half(); out = torch.zeros(1, C, device=dev)
    for _ in range(3): AD.colsum(dy, M, 1, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): AD.colsum(dy, M, 1, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"[{M:8d}, {C:5d}]: {us:7.1f} us  {M * C * 2 / us / 1e6:6.2f} TB/s")
