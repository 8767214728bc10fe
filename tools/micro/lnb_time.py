"""Time pt_layernorm_bwd on the training step's shapes (MI355X)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from posetraj_amd import hip, ops
dev = torch.device("cuda:0")
L = hip.lib()
for M, C in ((40320, 320), (10080, 640), (2520, 1280), (630, 1280), (2880, 320)):
    x = torch.randn(M, C, device=dev).half(); dy = torch.randn(M, C, device=dev).half(); dx = torch.empty_like(x)
    rs = torch.empty(2 * M, device=dev); g = torch.ones(C, device=dev).half(); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    for params in (False, True):
        def run():
            hip.check(L.pt_layernorm_bwd(x.data_ptr(), M, C, g.data_ptr(), 1e-5, dy.data_ptr(), dx.data_ptr(), dg.data_ptr() if params else None,
                                         db.data_ptr() if params else None, rs.data_ptr(), ops._stream()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"[{M:6d}, {C:5d}] params={params!s:5}: {us:7.1f} us  {3 * M * C * 2 / us / 1e6:6.2f} TB/s")
