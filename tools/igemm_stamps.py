#!/usr/bin/env python3
"""Where a tile's time goes in the pipelined igemm kernels (MI355X): s_memtime stamps per wave via pt_igemm_set_stamps.

    python tools/igemm_stamps.py M N K [cfg] [geglu] [res] [vec]     # cfg 0 = 256x256 (8-phase), 3 = 256x320
Prints the median cycles (100 MHz s_memtime ticks are converted with the measured kernel time) of: start -> first K
tile landed, main loop, epilogue, and the gap between consecutive tiles on one CU."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_linear

M, N, K = (int(v) for v in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
geglu = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
use_res = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
use_vec = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).half().to(dev)
w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
b = torch.randn(N, generator=g).half().to(dev)
pw = pack_linear(w, b, dev, geglu=geglu)
r = torch.randn(M, pw.n_out, generator=g).half().to(dev) if use_res else None
out = torch.empty(M, pw.n_out, dtype=torch.float16, device=dev)
vkw = dict(vec=torch.randn(2, pw.n_out, generator=g).half().to(dev), vec_mode=1, vG=M // 2) if use_vec else {}
bm, bn = 256, (256 if cfg == 0 else 320)
ntiles = -(-M // bm) * -(-N // bn)
stamps = torch.zeros(ntiles * 8 * 16, dtype=torch.int64, device=dev)
L = hip.lib()
hip.check(L.pt_igemm_force_config(cfg))
for _ in range(3):
    ops.igemm(x, pw, res=r, out=out, **vkw)
torch.cuda.synchronize()
hip.check(L.pt_igemm_set_stamps(stamps.data_ptr(), stamps.numel()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.igemm(x, pw, res=r, out=out, **vkw); e1.record(); torch.cuda.synchronize()
hip.check(L.pt_igemm_set_stamps(None, 0))
hip.check(L.pt_igemm_force_config(-1))
us = e0.elapsed_time(e1) * 1e3
s = stamps.cpu().numpy().reshape(ntiles, 8, 16).astype(np.float64)
# s_memtime bases differ between XCDs: only differences inside one wave mean anything.  Tick = shader cycle.
pro = np.median(s[:, :, 1] - s[:, :, 0])
loop = np.median(s[:, :, 2] - s[:, :, 1])
epi = np.median(s[:, :, 3] - s[:, :, 2])
tile = np.median(s[:, :, 3] - s[:, :, 0])
rounds = -(-ntiles // 256)
ghz = rounds * tile / (us * 1e3)                        # rough: kernel time ~ rounds x one tile's lifetime
print(f"M={M} N={N} K={K} cfg={cfg} geglu={int(geglu)} res={int(use_res)} vec={int(use_vec)}: kernel {us:.1f} us, {ntiles} tiles "
      f"({ntiles / 256:.2f} rounds), {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
print(f"  per wave, median cycles: prologue {pro:.0f}  main loop {loop:.0f} ({loop / (K // 64):.0f} per K tile)  "
      f"epilogue {epi:.0f}  whole tile {tile:.0f}   [rounds x tile / kernel time = {ghz:.2f} GHz-equivalent]")
bar = np.median(s[:, :, 4] - s[:, :, 2])
parts = [f"barrier {bar:.0f}"]
prev = 4
for c_ in range(4):
    if (s[:, :, 5 + 2 * c_] > 0).all():
        parts.append(f"chunk{c_}: stage {np.median(s[:, :, 5 + 2 * c_] - s[:, :, prev]):.0f} + rows {np.median(s[:, :, 6 + 2 * c_] - s[:, :, 5 + 2 * c_]):.0f}")
        prev = 6 + 2 * c_
print("  epilogue split (cycles): " + "; ".join(parts))
