#!/usr/bin/env python3
"""Where a tile of the spatial attention kernel spends its time: s_memtime stamps written by a STAMPED experimental build of
csrc/attn.hip (tools/variants/_build/libpt_attn_<v>s.so: workgroups 2048 .. 2559 of the level-0 launch record five program points
of tiles 60 .. 67 per wave; `pt_dbg_attn_stamps` copies them out).
    PT_LIB=tools/variants/_build/libpt_attn_v6s.so python3 tools/attn_stamps.py
slots: 0 tile start, 1 score MFMAs / V^T reads / next tile's copies issued, 2 the query's tile maximum known (score MFMAs done,
max chain, lane-pair exchange), 3 exps / converts / PV MFMAs issued, 4 behind the barrier; slot 7 of tile 0: XCC_ID << 32 | HW_ID."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posetraj_amd import hip
hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import ops
dev = torch.device("cuda:0")
Nimg, S, heads = 28, 9216, 5
qkv = torch.randn(Nimg * S, 3 * heads * 64, device=dev, dtype=torch.float16)
for _ in range(3):
    ops.attn_spatial(qkv, Nimg, S, heads, 64, q_prescaled=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(hip.LIB_PATH)
if os.environ.get("ATTN8"):        # the 8-wave ping-pong build: [512 WGs][8 waves][8 tiles][8 slots], slots 0 .. 6
    buf = np.zeros((512, 8, 8, 8), dtype=np.uint64)
    assert lib.pt_dbg_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf[..., :7].astype(np.int64)
    ok = (st > 0).all(axis=(2, 3))
    print(f"# {int(ok.sum())} of {ok.size} waves recorded; cycles (s_memtime), median [10 % .. 90 %] over waves and tiles 60 .. 67")
    names = ["VALU phase: 16 V^T reads + 2 copies issued", "            max chain + pair exchange", "            32 exp, 16 cvt, V^T wait",
             "            vmcnt(4) + barrier", "MFMA phase: 8 K reads + 20 MFMAs issued", "            vmcnt(2) + barrier"]
    for g, gname in ((slice(0, 4), "group A (waves 0-3)"), (slice(4, 8), "group B (waves 4-7)")):
        d = np.diff(st[:, g], axis=-1)[ok[:, g]]
        print(gname)
        for i, n in enumerate(names):
            v = d[..., i].ravel()
            print(f"  {n:46s} {np.median(v):7.0f}  [{np.percentile(v, 10):6.0f} .. {np.percentile(v, 90):6.0f}]")
        x = st[:, g][ok[:, g]]
        tile = (x[:, 1:, 0] - x[:, :-1, 0]).ravel()
        print(f"  {'tile period':46s} {np.median(tile):7.0f}  [{np.percentile(tile, 10):6.0f} .. {np.percentile(tile, 90):6.0f}]")
    # offset between the groups: start of B's VALU phase minus start of A's VALU phase of the same tile (same SIMD: waves w, w + 4)
    both = ok[:, :4] & ok[:, 4:]
    off = (st[:, 4:, :, 0] - st[:, :4, :, 0])[both].ravel()
    print(f"# B's VALU phase starts {np.median(off):.0f} cycles after A's [{np.percentile(off, 10):.0f} .. {np.percentile(off, 90):.0f}]")
    sys.exit(0)
if os.environ.get("ATTN_PIPE"):    # the 3-stage in-wave pipeline: slots 0 .. 5
    buf = np.zeros((512, 4, 8, 8), dtype=np.uint64)
    assert lib.pt_dbg_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf[..., :6].astype(np.int64)
    ok = (st > 0).all(axis=(2, 3))
    print(f"# {int(ok.sum())} of {ok.size} waves recorded; cycles (s_memtime), median [10 % .. 90 %] over waves and tiles 60 .. 67")
    names = ["vmcnt(4) + barrier", "4 copies + 8 K reads issued", "block: 20 MFMAs, 32 exp, 16 cvt, 16 V^T reads issued", "V^T wait (lgkmcnt(0))",
             "scores complete + max tree + pair exchange (+ re-base)"]
    d = np.diff(st, axis=-1)[ok]
    for i, n in enumerate(names):
        v = d[..., i].ravel()
        print(f"  {n:56s} {np.median(v):7.0f}  [{np.percentile(v, 10):6.0f} .. {np.percentile(v, 90):6.0f}]")
    tile = (st[ok][:, 1:, 0] - st[ok][:, :-1, 0]).ravel()
    print(f"  {'tile period':56s} {np.median(tile):7.0f}  [{np.percentile(tile, 10):6.0f} .. {np.percentile(tile, 90):6.0f}]")
    sys.exit(0)
buf = np.zeros((512, 4, 8, 8), dtype=np.uint64)
rc = lib.pt_dbg_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
st = buf[..., :5].astype(np.int64)
ok = (st > 0).all(axis=(2, 3))
print(f"# {int(ok.sum())} of {ok.size} waves recorded; cycles (s_memtime), median [10 % .. 90 %] over waves and tiles 60 .. 67")
d = np.diff(st, axis=-1)[ok]                                   # [waves, 8 tiles, 4 phases]
names = ["issue: K reads, 8 score MFMAs, 16 V^T reads, 4 copies", "score MFMAs complete + max chain + pair exchange",
         "32 exp, 16 cvt, 8 PV + 4 row-sum MFMAs issued", "barrier (incl. vmcnt(0) for the next tile's copies)"]
for i, n in enumerate(names):
    v = d[..., i].ravel()
    print(f"  {n:58s} {np.median(v):7.0f}  [{np.percentile(v, 10):6.0f} .. {np.percentile(v, 90):6.0f}]")
tile = (st[ok][:, 1:, 0] - st[ok][:, :-1, 0]).ravel()
print(f"  {'tile period (start to start)':58s} {np.median(tile):7.0f}  [{np.percentile(tile, 10):6.0f} .. {np.percentile(tile, 90):6.0f}]")
# how the two waves of a SIMD (one per workgroup) sit against each other: phase offset of co-resident waves
hw = buf[:, :, 0, 7]
cu_simd = {}
for b in range(512):
    for w in range(4):
        if not ok[b, w]:
            continue
        h = int(hw[b, w]); xcc = h >> 32; hid = h & 0xFFFFFFFF
        key = (xcc, (hid >> 8) & 0xF, (hid >> 13) & 0x7, (hid >> 4) & 0x3)      # XCC, CU, SE, SIMD
        cu_simd.setdefault(key, []).append((b, w))
offs = []
for key, lst in cu_simd.items():
    if len(lst) == 2:
        (b0, w0), (b1, w1) = lst
        a = st[b0, w0, :, 0]; c = st[b1, w1, :, 0]
        per = np.median(np.diff(a))
        if abs(int(a[0]) - int(c[0])) < 20 * per:
            offs.append(((c[3] - a[3]) % per) / per)
if offs:
    h, _ = np.histogram(offs, bins=10, range=(0, 1))
    print(f"# phase of the SIMD's second wave inside the first one's tile period, {len(offs)} SIMDs: histogram over tenths {h.tolist()}")
