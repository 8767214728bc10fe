#!/usr/bin/env python3
"""A few launches of the level-0 GEGLU feed-forward in both forms (two pt_igemm_f16 launches / pt_ffn_geglu_f16) for a profiler:
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d <dir> -o p -- python3 tools/ffn_one.py
    python3 tools/pmc_summary.py <dir>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posetraj_amd import ops
from posetraj_amd.packing import pack_linear
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, C, I = 258048, 320, 1280
r16 = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().to(dev)
x, res = r16(M, C), r16(M, C)
p1 = pack_linear(r16(2 * I, C, sc=C ** -0.5), r16(2 * I, sc=0.3), dev, geglu=True)
p2 = pack_linear(r16(C, I, sc=I ** -0.5), r16(C, sc=0.3), dev)
out = torch.empty(M, C, dtype=torch.float16, device=dev)
mid = torch.empty(M, I, dtype=torch.float16, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.igemm(ops.igemm(x, p1, out=mid), p2, res=res, out=out)
    ops.ffn_geglu(x, p1, p2, res=res, out=out)
torch.cuda.synchronize()
print("done")
