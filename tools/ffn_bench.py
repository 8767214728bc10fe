#!/usr/bin/env python3
"""pt_ffn_geglu_f16 (one launch) against the two pt_igemm_f16 launches it replaces, alone on the device, on the level-0 shape of the
headline workload (M = 28 x 72 x 128 = 258 048 rows, C = 320, inner = 1280), N(0,1)-scaled random operands, interleaved rounds
in ONE process (guide rule 24), with socket power and the in-kernel clock of each arm (tools/energy_table.py's samplers).
    python tools/ffn_bench.py [--rows 258048] [--seconds 2.0] [--rounds 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from posetraj_amd import ops
from posetraj_amd.packing import pack_linear
import energy_table as ET

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=258048); ap.add_argument("--seconds", type=float, default=2.0); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--gelu-modes", nargs="*", default=[], help="(round-5 tuning builds only, tools/variants: PT_FFN_GELU = 1 scalar GELU polynomial, 2 no GELU)")
ap.add_argument("--cases", nargs="*", default=None)
ap.add_argument("--with-pre", action="store_true", help="time the feed-forward WITH the attention output projection and LayerNorm in front of it: three launches vs one (pre=)")
ap.add_argument("--variant-b", action="store_true", help="(with tools/variants/ffn_variant_b_gelu_spread.diff built in) also time ffn320b_kernel, PT_FFN_V=b")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, C, I = a.rows, 320, 1280
r16 = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().to(dev)
x, res, blend = r16(M, C), r16(M, C), r16(M, C)
vec = r16(28, C)
p1 = pack_linear(r16(2 * I, C, sc=C ** -0.5), r16(2 * I, sc=0.3), dev, geglu=True)
p2 = pack_linear(r16(C, I, sc=I ** -0.5), r16(C, sc=0.3), dev)
out = torch.empty(M, C, dtype=torch.float16, device=dev)
mid = torch.empty(M, I, dtype=torch.float16, device=dev)
flops = 2.0 * M * (2 * I) * C + 2.0 * M * C * I
cases = {"res": dict(res=res), "res+vec": dict(res=res, vec=vec, vec_mode=1, vG=M // 28), "res+blend": dict(res=res, blend=blend, alpha=0.4)}


def arm(fn, seconds):
    fn(); torch.cuda.synchronize()
    clock = ET.Clock(seconds); clock.start()
    pw = ET.Power(0).start()
    cur = torch.cuda.current_stream()                           # stream-level waits only: a device-wide synchronize waits for the probe
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); n = 0
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        cur.synchronize()
    e1.record(); e1.synchronize()
    t1 = time.perf_counter()
    us = 1000 * e0.elapsed_time(e1) / n
    w, _ = pw.mean_between(t0 + seconds / 3, t1)
    ck = clock.finish(seconds / 3, t1 - t0)
    return us, w, ck


print(f"# fused GEGLU feed-forward vs two launches, M = {M}, C = {C}, inner = {I}: {flops / 1e9:.1f} GFLOP per feed-forward; {torch.cuda.get_device_name(0)}")
if a.with_pre:      # out-projection + residual + row vector, LayerNorm, feed-forward:  3 launches  vs  ONE (pre=)
    att = r16(M, C)
    po = pack_linear(r16(C, C, sc=C ** -0.5), r16(C, sc=0.2), dev)
    gam, bet = (1.0 + 0.2 * torch.randn(C, generator=g)).half().to(dev), (0.1 * torch.randn(C, generator=g)).half().to(dev)
    xv = r16(2, C)
    hbuf, ybuf = torch.empty_like(x), torch.empty_like(x)
    fl = flops + 2.0 * M * C * C
    for name, bkw in (("spatial ff", {}), ("temporal ff", dict(blend=blend, alpha=0.4))):
        vkw = dict(vec=xv, vec_mode=1, vG=M // 2) if name == "spatial ff" else dict(vec=xv, vec_mode=2, vFS=M // 2, vS=M // 28, vB=2)
        def comp():
            hh = ops.igemm(att, po, res=res, out=hbuf, **vkw)
            return ops.ffn_geglu(ops.layernorm(hh, gam, bet), p1, p2, res=hh, out=out, **bkw)
        fused = lambda: ops.ffn_geglu(att, p1, p2, out=out, pre=dict(w=po, res=res, ln=(gam, bet, 1e-5), **vkw), **bkw)
        r_ = comp().clone(); g_ = fused().clone(); torch.cuda.synchronize()
        print(f"{name:12s} fused vs composition: rel-L2 {float((g_.float() - r_.float()).norm() / r_.float().norm()):.2e}")
        rows = {"3 launches (out-proj, LN, fused ff)": [], "1 launch (pre=)": []}
        for r in range(a.rounds):
            for label, fn in (("3 launches (out-proj, LN, fused ff)", comp), ("1 launch (pre=)", fused)):
                rows[label].append(arm(fn, a.seconds))
        for label, v in rows.items():
            us = min(x_[0] for x_ in v); w = sum(x_[1] or 0 for x_ in v) / len(v)
            print(f"{name:12s} {label:36s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s   {w:6.0f} W   {w * us * 1e-6:6.3f} J   (rounds: {', '.join(f'{x_[0]:.1f}' for x_ in v)})")
    sys.exit(0)
for name, kw in cases.items():
    if a.cases and name not in a.cases:
        continue
    two = lambda: ops.igemm(ops.igemm(x, p1, out=mid), p2, out=out, **kw)
    one = lambda: ops.ffn_geglu(x, p1, p2, out=out, **kw)
    ref = two().clone(); got = one().clone(); torch.cuda.synchronize()
    same = torch.equal(ref, got)
    arms = [("two launches", two, None), ("fused", one, "0")] + [(f"fused PT_FFN_GELU={m}", one, m) for m in a.gelu_modes]
    if a.variant_b:
        arms.append(("fused variant B", one, "b"))
        os.environ["PT_FFN_V"] = "b"; gb = one().clone(); cur_sync = torch.cuda.synchronize(); os.environ.pop("PT_FFN_V")
        d = (gb.float() - ref.float())
        print(f"{name:10s} variant B vs two launches: rel-L2 {float(d.norm() / ref.float().norm()):.2e}, max |d| {float(d.abs().max()):.2e}, "
              f"{100.0 * float((gb != ref).float().mean()):.2f} % of the values differ")
    rows = {l: [] for l, _, _ in arms}
    for r in range(a.rounds):
        for label, fn, mode in arms:
            if mode == "b":
                os.environ["PT_FFN_V"] = "b"
            elif mode is not None:
                os.environ["PT_FFN_GELU"] = mode
            rows[label].append(arm(fn, a.seconds))
            os.environ.pop("PT_FFN_GELU", None); os.environ.pop("PT_FFN_V", None)
    for label, v in rows.items():
        us = min(x_[0] for x_ in v)
        w = sum(x_[1] or 0 for x_ in v) / len(v)
        ck = [x_[2][0] for x_ in v if x_[2]]
        print(f"{name:10s} {label:13s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   {w:6.0f} W   {(sum(ck) / len(ck) if ck else 0):.2f} GHz   "
              f"{w * us * 1e-6:6.3f} J   (rounds: {', '.join(f'{x_[0]:.1f}' for x_ in v)})" + ("" if same else "   RESULTS DIFFER"))
