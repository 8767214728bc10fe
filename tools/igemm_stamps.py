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
conv3 = bool(int(sys.argv[8])) if len(sys.argv) > 8 else False    # 1: 3x3 convolution over 28 x 72 x 128 pixels, K = 9 C (M must be 258048)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).half().to(dev)
w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
b = torch.randn(N, generator=g).half().to(dev)
pw = pack_linear(w, b, dev, geglu=geglu)
geom = None
if conv3:
    from posetraj_amd.packing import pack_conv2d
    assert M == 28 * 72 * 128 and K % 9 == 0
    x = torch.randn(M, K // 9, generator=g).half().to(dev).view(28, 72, 128, K // 9)
    pw = pack_conv2d(w.view(N, 3, 3, K // 9).permute(0, 3, 1, 2).contiguous(), b, dev)
    geom = (28, 72, 128)
r = torch.randn(M, pw.n_out, generator=g).half().to(dev) if use_res else None
out = torch.empty(M, pw.n_out, dtype=torch.float16, device=dev)
vkw = dict(vec=torch.randn(2, pw.n_out, generator=g).half().to(dev), vec_mode=1, vG=M // 2) if use_vec else {}
bm, bn = 256, (256 if cfg == 0 else 320)
ntiles = -(-M // bm) * -(-N // bn)
stamps = torch.zeros(ntiles * 8 * 16, dtype=torch.int64, device=dev)
L = hip.lib()
hip.check(L.pt_igemm_force_config(cfg))
for _ in range(3):
    ops.igemm(x, pw, res=r, out=out, **vkw, **({'geom': geom} if geom else {}))
torch.cuda.synchronize()
hip.check(L.pt_igemm_set_stamps(stamps.data_ptr(), stamps.numel()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.igemm(x, pw, res=r, out=out, **vkw, **({'geom': geom} if geom else {})); e1.record(); torch.cuda.synchronize()
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

# ---- per-CU timelines (slot 15 = XCC_ID << 32 | HW_ID): how much of a CU's time lies between its workgroups
hw = s[:, 0, 15].astype(np.int64)
cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 8) & 0xFF)                # (xcc, se/sh/cu bits 8..15 of HW_ID)
start = s[:, :, 0].min(axis=1); end = s[:, :, 3].max(axis=1)           # workgroup = first wave in .. last wave out
wave_skew_out = np.median(s[:, :, 3].max(axis=1) - s[:, :, 3].min(axis=1))
wave_skew_in = np.median(s[:, :, 0].max(axis=1) - s[:, :, 0].min(axis=1))
busy, span, gaps, per_round = [], [], [], {}
for k in np.unique(cu_key):
    idx = np.where(cu_key == k)[0]
    o = idx[np.argsort(start[idx])]
    busy.append((end[o] - start[o]).sum()); span.append(end[o].max() - start[o].min())
    gaps.extend((start[o][1:] - end[o][:-1]).tolist())
    for r_, t_ in enumerate(o):
        per_round.setdefault(r_, []).append(end[t_] - start[t_])
busy, span = np.array(busy), np.array(span)
print(f"  {len(busy)} CUs seen; workgroup lifetime (first wave in .. last wave out) median {np.median(end - start):.0f} cycles, "
      f"waves enter within {wave_skew_in:.0f} and leave within {wave_skew_out:.0f} cycles of each other")
print(f"  per CU: span median {np.median(span):.0f} cycles -> {np.median(span) / (us * 1e3):.2f} GHz if the span is the kernel; "
      f"busy/span median {np.median(busy / span):.3f}; gap between workgroups median {np.median(gaps) if gaps else 0:.0f} "
      f"p90 {np.percentile(gaps, 90) if gaps else 0:.0f} cycles")
print("  workgroup lifetime by position on its CU: " + "  ".join(f"#{r_}: {np.median(v):.0f}" for r_, v in sorted(per_round.items())))
