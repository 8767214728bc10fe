#!/usr/bin/env python3
"""Times pt_igemm_f16 on the short-K / small-M shapes of the 14x576x1024 workload (hipEvents, median of N launches).

    python tools/igemm_bench.py [--reps 20] [--cfg -1] [SHAPE ...]      SHAPE = M,N,K[,geglu[,res[,vec[,wide]]]]
Library switches are read once per process (PT_IGEMM_*): run once per variant, e.g. under PT_LIB=<other .so>."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip
if os.environ.get("PT_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import ops
from posetraj_amd.packing import pack_linear

DEFAULT = ["258048,2560,320,1", "258048,960,320", "258048,320,320,0,1,1", "258048,320,320,0,1", "258048,320,320",
           "258048,320,1280,0,1", "64512,5120,640,1", "64512,1920,640", "64512,640,640,0,1,1", "64512,640,2560,0,1",
           "16128,10240,1280,1", "16128,1280,1280,0,1,1", "16128,1280,5120,0,1"]
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--cfg", type=int, default=-1)
ap.add_argument("shapes", nargs="*")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
hip.check(hip.lib().pt_igemm_force_config(a.cfg))
tot = 0.0
for sh in (a.shapes or DEFAULT):
    v = [int(t) for t in sh.split(",")] + [0, 0, 0, 0]
    M, N, K, geglu, use_res, use_vec, wide = v[:7]
    x = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    b = torch.randn(N, generator=g).half().to(dev)
    pw = pack_linear(w, b, dev, geglu=bool(geglu))
    r = torch.randn(M, pw.n_out, generator=g).half().to(dev) if use_res else None
    if r is not None and wide:
        r.lo = (torch.randn(M, pw.n_out, generator=g) * 2.0 ** -12).half().to(dev)
    out = torch.empty(M, pw.n_out, dtype=torch.float16, device=dev)
    vkw = dict(vec=torch.randn(2, pw.n_out, generator=g).half().to(dev), vec_mode=1, vG=M // 2) if use_vec else {}
    for _ in range(3):
        ops.igemm(x, pw, res=r, out=out, wide=bool(wide), **vkw)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for e0, e1 in ev:
        e0.record(); ops.igemm(x, pw, res=r, out=out, wide=bool(wide), **vkw); e1.record()
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)[a.reps // 2]
    tot += us
    print(f"{M:7d} {N:6d} {K:6d} geglu={geglu} res={use_res} vec={use_vec} wide={wide}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    del x, w, pw, r, out
print(f"sum {tot:.1f} us")
