#!/usr/bin/env python3
"""The level-0 launch of the spatial attention kernel run back to back for a few seconds, with socket power and the in-kernel
clock (tools/energy_table.py's samplers): time, TFLOP/s, W, GHz, J per launch and core cycles per 64-key tile and wave.
PT_LIB=<path> loads an experimental build (tools/variants/_build/libpt_attn_*.so) instead of the in-tree library.
    python tools/attn_bench.py [--seconds 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from posetraj_amd import hip
if os.environ.get("PT_LIB"):
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import ops
import energy_table as ET
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=3.0); a = ap.parse_args()
dev = torch.device("cuda:0")
Nimg, S, heads = 28, 9216, 5
qkv = torch.randn(Nimg * S, 3 * heads * 64, device=dev, dtype=torch.float16)
fn = lambda: ops.attn_spatial(qkv, Nimg, S, heads, 64, q_prescaled=True)
for _ in range(5):
    fn()
torch.cuda.synchronize()
clock = ET.Clock(a.seconds); clock.start()
pw = ET.Power(0).start()
cur = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); n = 0
e0.record()
while time.perf_counter() - t0 < a.seconds:
    for _ in range(20):
        fn()
    n += 20
    cur.synchronize()
e1.record(); e1.synchronize()
t1 = time.perf_counter()
ms = e0.elapsed_time(e1) / n
w, _ = pw.mean_between(t0 + a.seconds / 3, t1)
ck = clock.finish(a.seconds / 3, t1 - t0)
ghz = ck[0] if ck else float("nan")
tiles = (Nimg * heads * ((S + 127) // 128)) / 512.0 * (S // 64)      # tile periods a resident workgroup slot runs through
print(f"{os.environ.get('PT_LIB', 'in-tree'):48s} {ms:7.3f} ms  {4.0 * Nimg * heads * S * S * 64 / ms / 1e9:6.0f} TFLOP/s  {w or 0:5.0f} W  {ghz:.2f} GHz  "
      f"{(w or 0) * ms * 1e-3:6.3f} J  {ms * 1e-3 * ghz * 1e9 / tiles:6.0f} cycles per tile and wave")
