#!/usr/bin/env python3
"""Times the implicit-GEMM kernel's tile configurations on the path's dominant linear shapes (MI355X).
    python tools/igemm_cfg_sweep.py > profiles/igemm_cfg_sweep_<tag>.txt
    SWEEP_WORKLOAD=M python tools/igemm_cfg_sweep.py        the same layer shapes at BASELINE configs[1]'s row counts (14 x 320 x 576: 80 640 /
                                                             20 160 / 5 040 / 1 260 rows instead of 258 048 / 64 512 / 16 128 / 4 032)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip, ops
if os.environ.get("PT_LIB"):                       # experimental build of the library (tuning sessions)
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd.packing import pack_linear

dev = torch.device("cuda:0")
# M, N, K, geglu, epilogue(res): the dominant 1x1 / linear shapes of workload L plus im2col-equivalent K extents
SHAPES = [  # M, N, K, geglu, epilogue(res)
    (258048, 2560, 320, True, False), (258048, 960, 320, False, False), (258048, 320, 320, False, True),
    (258048, 320, 1280, False, True), (64512, 5120, 640, True, False), (64512, 1920, 640, False, False),
    (64512, 640, 640, False, True), (64512, 640, 2560, False, True), (16128, 10240, 1280, True, False),
    (16128, 3840, 1280, False, False), (16128, 1280, 1280, False, True), (16128, 1280, 5120, False, True),
    (4032, 1280, 11520, False, True), (4032, 10240, 1280, True, False), (4032, 1280, 5120, False, True),
    (258048, 320, 2880, False, True), (64512, 640, 5760, False, True), (16128, 1280, 11520, False, True),
    (258048, 320, 960, False, True), (64512, 640, 1920, False, True), (16128, 1280, 3840, False, True),
    (16128, 1280, 23040, False, True), (64512, 640, 11520, False, True), (258048, 320, 5760, False, True),
    # round 6: every level-3 launch family and the short-K level-2 ones (VERDICT r05 #1a)
    (4032, 1280, 3840, False, True), (4032, 1280, 1280, False, True), (4032, 3840, 1280, False, False), (4032, 1280, 23040, False, True),
    (4032, 1280, 2560, False, True), (16128, 1280, 2560, False, True),
]
if os.environ.get("SWEEP_WORKLOAD", "L") == "M":
    rows = {258048: 80640, 64512: 20160, 16128: 5040, 4032: 1260}
    SHAPES = [(rows[M],) + tuple(r) for M, *r in SHAPES]
if os.environ.get("SWEEP_ONLY"):                   # comma-separated indices into SHAPES
    SHAPES = [SHAPES[int(i)] for i in os.environ["SWEEP_ONLY"].split(",")]
NAMES = {0: "256x256", 1: "128x320", 2: "128x128", 3: "256x320", 4: "128x160"}
g = torch.Generator().manual_seed(0)
CFGS = (0, 1, 2, 3, 4)                             # (5 = 256 x 32 serves N <= 96 only; the two-per-CU 256 x 160 kernel of round 3 is gone)
print(f"{'M':>7} {'N':>6} {'K':>6} g r | " + " | ".join(f"{NAMES[c]:>16}" for c in CFGS) + " | auto")
for M, N, K, geglu, res in SHAPES:
    x = (torch.randn(M, K, generator=g)).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    b = torch.randn(N, generator=g).half().to(dev)
    pw = pack_linear(w, b, dev, geglu=geglu)
    r = torch.randn(M, pw.n_out, generator=g).half().to(dev) if res else None
    vkw = {}
    if os.environ.get("SWEEP_VEC") and res:            # second side input: a per-clip row vector (epilogue code 3)
        vkw = dict(vec=torch.randn(2, pw.n_out, generator=g).half().to(dev), vec_mode=1, vG=M // 2)
    out = torch.empty(M, pw.n_out, dtype=torch.float16, device=dev)
    cfgs = [c for c in CFGS + (-1,) if not (c == 1 and geglu)]
    times = {c: [] for c in cfgs}
    for c in cfgs:                                   # warm-up (clocks, caches) before any timing
        hip.check(hip.lib().pt_igemm_force_config(c))
        for _ in range(5):
            ops.igemm(x, pw, res=r, out=out, **vkw)
    torch.cuda.synchronize()
    for rnd in range(5):                             # interleaved rounds: one process, one device (guide rule 24)
        for c in cfgs:
            hip.check(hip.lib().pt_igemm_force_config(c))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.igemm(x, pw, res=r, out=out, **vkw)
            e1.record(); torch.cuda.synchronize()
            times[c].append(e0.elapsed_time(e1) * 200)
    hip.check(hip.lib().pt_igemm_force_config(-1))
    cells = []
    for c in CFGS + (-1,):
        if c not in times:
            cells.append(f"{'-':>16}")
        else:
            us = sorted(times[c])[len(times[c]) // 2]
            cells.append(f"{us:8.1f}us {2.0 * M * N * K / us / 1e6:5.0f}T")
    print(f"{M:7d} {N:6d} {K:6d} {int(geglu)} {int(res)} | " + " | ".join(cells))
