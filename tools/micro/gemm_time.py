"""pt_gemm_f16 on the weight-gradient shapes of one training step (14 x 320 x 576; profiles/r04/train_step_bench_r04.txt):
both operands with their unit stride across the rows (dY^T and X), fp32 result.  PT_LIB selects the library.

    python tools/micro/gemm_time.py [--iters 30]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from posetraj_amd import autodiff as AD

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30); a = ap.parse_args()
dev = torch.device("cuda:0")
F = 14
SHAPES = [  # M, N, K(pixels), taps, splits, out_mode, (H, W) of the level or None
    (10240, 1280, 2520, 1, 1, 3, None), (2560, 320, 40320, 1, 17, 2, None), (5120, 640, 10080, 1, 5, 2, None),
    (320, 320, 40320, 9, 12, 2, (40, 72)), (1280, 1280, 630, 9, 1, 3, (5, 9)), (640, 2560, 10080, 1, 10, 2, None),
    (1280, 5120, 2520, 1, 2, 2, None), (320, 320, 40320, 1, 78, 2, None), (320, 1280, 40320, 1, 34, 2, None),
    (1280, 1280, 2520, 9, 1, 3, (10, 18)), (640, 640, 10080, 9, 4, 2, (20, 36)), (640, 640, 10080, 1, 19, 2, None),
    (1280, 1280, 2520, 1, 4, 2, None), (1280, 1280, 630, 1, 1, 3, None),
]
g = torch.Generator(device="cpu").manual_seed(0)
tot = 0.0
for M, N, K, T, splits, mode, hw in SHAPES:
    dy = (torch.randn(K, M, generator=g) * 0.1).half().to(dev)
    x = (torch.randn(K, N, generator=g) * 0.1).half().to(dev)
    out = torch.zeros(T, M, N, dtype=torch.float32, device=dev)
    gather = None if hw is None else (hw[0], hw[1], hw[0], hw[1], 3, 3, 1, 1, 1, N)
    def run():
        AD.gemm((dy, 0), (x, 0), (out, 0), M, N, K, (1, M), (N, 1), (N, 1), nb=(1, 1, T), bc=(0, 0, M * N), out_mode=mode, splits=splits, gather=gather)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 2.0 * M * N * K * T
    tot += us
    print(f"{M:6d} x {N:5d} x {K:6d} taps {T} splits {splits:3d} mode {mode} {'gather' if hw else '      '} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s")
print(f"sum {tot:.1f} us")
