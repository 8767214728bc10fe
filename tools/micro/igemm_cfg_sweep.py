"""Tile-configuration sweep of pt_igemm_f16 on the small-M shapes of the training step (14 x 320 x 576: levels 2-3 have 2520 / 630
rows; the one-frame spatial pass 180 / 45) - the automatic choice was tuned on the inference workloads' shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_conv2d, pack_linear
dev = torch.device("cuda:0")
L = hip.lib()
g = torch.Generator().manual_seed(0)
shapes = [("lin", 2520, 1280, 1280), ("lin", 630, 1280, 1280), ("lin", 180, 1280, 1280), ("lin", 2520, 1280, 5120), ("lin", 2520, 10240, 1280), ("lin", 2520, 3840, 1280),
          ("lin", 10080, 640, 640), ("lin", 40320, 320, 320), ("lin", 720, 640, 2560),
          ("c33", 630, 1280, 1280), ("c33", 2520, 1280, 1280), ("c33", 180, 1280, 1280), ("c33", 2520, 1280, 2560), ("c33", 630, 1280, 2560)]
for kind, M, N, C in shapes:
    if kind == "lin":
        pw = pack_linear(torch.randn(N, C, generator=g) * C ** -0.5, torch.zeros(N), dev)
        x = torch.randn(M, C, device=dev).half()
        geom = None
    else:
        pw = pack_conv2d(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(N), dev)
        hw = {630: (5, 9), 2520: (10, 18), 180: (10, 18)}[M]
        n = M // (hw[0] * hw[1])
        x = torch.randn(n, hw[0], hw[1], C, device=dev).half()
        geom = (n, hw[0], hw[1])
    res = []
    for cfg in (-1, 3, 1, 2, 4):
        L.pt_igemm_force_config(cfg)
        try:
            for _ in range(3): ops.igemm(x, pw, geom=geom)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.igemm(x, pw, geom=geom)
            e1.record(); torch.cuda.synchronize()
            res.append(f"{'auto' if cfg < 0 else cfg}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
        except RuntimeError as e:
            res.append(f"{cfg}: n/a")
    L.pt_igemm_force_config(-1)
    fl = 2.0 * M * N * pw.K
    print(f"{kind} {M:6d} x {N:5d} x {pw.K:5d}  " + "   ".join(res) + f"   (1 PF = {fl / 1e9:.1f} us)")
