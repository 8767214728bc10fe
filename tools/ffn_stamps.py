#!/usr/bin/env python3
"""Where a chunk of pt_ffn_geglu_f16 spends its cycles (MI355X): the stamped build of the residual-only variant (ffn.hip, ST)
writes s_memtime at the phase boundaries of chunk 2 of every wave (pt_igemm_set_stamps).  Medians over all waves, early group
(waves 0-3) and late group (waves 4-7) apart.      python tools/ffn_stamps.py [rows]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_linear

M = int(sys.argv[1]) if len(sys.argv) > 1 else 258048
C, I = 320, 1280
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r16 = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().to(dev)
x, res = r16(M, C), r16(M, C)
p1 = pack_linear(r16(2 * I, C, sc=C ** -0.5), r16(2 * I, sc=0.3), dev, geglu=True)
p2 = pack_linear(r16(C, I, sc=I ** -0.5), r16(C, sc=0.3), dev)
out = torch.empty(M, C, dtype=torch.float16, device=dev)
nwg = -(-M // 128)
stamps = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device=dev)
L = hip.lib()
for _ in range(3):
    ops.ffn_geglu(x, p1, p2, res=res, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.ffn_geglu(x, p1, p2, res=res, out=out); e1.record(); torch.cuda.synchronize()
us_plain = e0.elapsed_time(e1) * 1e3
hip.check(L.pt_igemm_set_stamps(stamps.data_ptr(), stamps.numel()))
e0.record(); ops.ffn_geglu(x, p1, p2, res=res, out=out); e1.record(); torch.cuda.synchronize()
hip.check(L.pt_igemm_set_stamps(None, 0))
us = e0.elapsed_time(e1) * 1e3
s = stamps.cpu().numpy().reshape(nwg, 8, 16).astype(np.float64)
# slots 4 .. 8 belong to the shared epilogue (igemm_tail.h: 4 = epilogue barrier passed, 5 + 2c / 6 + 2c = row chunk c staged / stored)
seq = [(0, "start"), (1, "prologue done"), (2, "chunk 2: P0 begin"), (3, "P0 issue part done (bias, 8 reads, 2 copies, vmcnt)"), (9, "P0 first barrier passed"),
       (10, "P0 end (16 MFMAs, second barrier)"), (11, "P4 end"), (12, "P5 issue part done (GELU, h stores, 8 reads, 2 copies)"), (13, "P5 end (barrier, h reads, 16 MFMAs, barrier)"),
       (14, "P7 end = chunk end")]
print(f"pt_ffn_geglu_f16 M={M}: {us_plain:.1f} us plain, {us:.1f} us with stamps; {nwg} workgroups ({nwg / 256:.2f} rounds)")
for grp, sl in (("early group (waves 0-3)", slice(0, 4)), ("late group (waves 4-7)", slice(4, 8))):
    v = s[:, sl, :]
    print(f"  {grp}: median cycles between consecutive stamps")
    for (i0, n0), (i1, n1) in zip(seq[:-1], seq[1:]):
        print(f"    {n0:60s} -> {n1:60s} {np.median(v[:, :, i1] - v[:, :, i0]):9.0f}")
    print(f"    chunk 2: {np.median(v[:, :, 14] - v[:, :, 2]):.0f} cycles; start -> epilogue barrier {np.median(v[:, :, 4] - v[:, :, 0]):.0f}; "
          f"epilogue (2 row chunks) {np.median(v[:, :, 8] - v[:, :, 4]):.0f}")
