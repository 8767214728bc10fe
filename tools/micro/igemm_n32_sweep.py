"""pt_igemm_f16 on the condition encoder's narrow convolutions (N <= 32 over up to 2.6 M pixels): 256 x 32 tiles (cfg 5, the automatic
choice for N <= 32) against the 128-wide ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_conv2d
dev = torch.device("cuda:0"); L = hip.lib()
g = torch.Generator().manual_seed(0)
for n, H, W, Ci, Co, stride in ((14, 320, 576, 8, 16, 1), (14, 320, 576, 16, 16, 1), (14, 320, 576, 16, 32, 2), (14, 160, 288, 32, 32, 1), (14, 160, 288, 32, 16, 1),
                                (14, 160, 288, 32, 96, 2), (2, 64, 64, 16, 16, 1)):
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5
    pw = pack_conv2d(w, torch.randn(Co, generator=g) * 0.1, dev, stride=stride)
    x = torch.randn(n, H, W, Ci, device=dev).half()
    want = None
    res = []
    for cfg in (-1, 5, 2, 4):
        L.pt_igemm_force_config(cfg)
        y = ops.igemm(x, pw, geom=(n, H, W))
        if want is None:
            want = torch.nn.functional.conv2d(x[:1].float().permute(0, 3, 1, 2).cpu(), w.half().float(), pw.bias[:Co].float().cpu(), stride=stride, padding=1)
        oh, ow = want.shape[2], want.shape[3]
        err = float((y.view(n, oh, ow, Co)[:1].float().cpu().permute(0, 3, 1, 2) - want).norm() / want.norm())
        for _ in range(2): ops.igemm(x, pw, geom=(n, H, W))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.igemm(x, pw, geom=(n, H, W))
        e1.record(); torch.cuda.synchronize()
        res.append(f"{'auto' if cfg < 0 else cfg}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us (err {err:.1e})")
    L.pt_igemm_force_config(-1)
    print(f"{n} x {H} x {W}  {Ci:3d} -> {Co:3d} s{stride}   " + "   ".join(res))
