#!/usr/bin/env python3
"""pt_ln_linear_f16 (LayerNorm + Q | K | V projection of the 320-channel level in one launch) against the two launches it replaces,
alone on the device, interleaved rounds in one process (MI355X).
    python tools/lnlin_bench.py [--rows 258048] [--n 960] > profiles/r06/lnlin_bench_alone.txt"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip, ops
if os.environ.get("PT_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd.packing import pack_linear

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, nargs="*", default=[258048, 80640]); ap.add_argument("--n", type=int, default=960)
a = ap.parse_args()
if os.environ.get("PT_LNLIN_DBG"):                 # an ablation instance of the kernel (results are wrong): profiles/r06/lnlin_ablations.txt
    hip.check(hip.lib().pt_ln_linear_set_ablation(int(os.environ["PT_LNLIN_DBG"])), "pt_ln_linear_set_ablation")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
K, N = 320, a.n
for M in a.rows:
    x = torch.randn(M, K, generator=g).half().to(dev)
    pw = pack_linear(torch.randn(N, K, generator=g) * K ** -0.5, None, dev)
    gam, bet = (1 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    y = torch.empty_like(x)
    forms = {
        "layernorm + igemm (two launches)": lambda: ops.igemm(ops.layernorm(x, gam, bet), pw, cs_cols=320, cs_scale=0.18, out=out),
        "igemm alone (the projection)": lambda: ops.igemm(x, pw, cs_cols=320, cs_scale=0.18, out=out),
        "ln_linear (one launch)": lambda: ops.ln_linear(x, gam, bet, pw, cs_cols=320, cs_scale=0.18, out=out),
    }
    times = {k: [] for k in forms}
    for f in forms.values():
        for _ in range(5):
            f()
    torch.cuda.synchronize()
    for rnd in range(7):
        for k, f in forms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 100)
    print(f"# {M} x {N} x {K}, median of 7 interleaved rounds of 10 launches")
    for k, v in times.items():
        us = sorted(v)[len(v) // 2]
        print(f"  {k:36s} {us:8.1f} us   {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s   (min {min(v):.1f}, max {max(v):.1f})")
