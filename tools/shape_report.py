#!/usr/bin/env python3
"""Per-shape efficiency table of the implicit-GEMM kernel over one denoise iteration (MI355X).
    python tools/shape_report.py [--workload L|M|S] > profiles/igemm_shapes_<tag>.txt"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from posetraj_amd import hip
if os.environ.get("PT_LIB"):                       # experimental build of the library (A/B on one box)
    hip.LIB_PATH = os.path.abspath(os.environ["PT_LIB"])
from posetraj_amd import (ControlNetSDVModel, EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet,
                          SVD_SCHEDULER_CONFIG, UNetSpatioTemporalConditionControlNetModel, ops)

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="L"); a = ap.parse_args()
H, W = bench.WORKLOADS[a.workload]
dev = torch.device("cuda:0")
unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=100, device=dev)
cn = ControlNetSDVModel(**bench.SVD).init_random_(seed=200, device=dev)
sched = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
pipe = StableVideoDiffusionPipelineControlNet(unet=unet, controlnet=cn, scheduler=sched)
sched.set_timesteps(2)
clip = bench.synth_clip(H, W, 14, 1024, 1234, dev, sched.init_noise_sigma)
pipe.denoise(*clip, num_inference_steps=2)                                   # warm-up (and condition-encoder cache)
torch.cuda.synchronize()
ops.Profiler.shapes = []
with ops.Profiler():
    pipe.denoise(*clip, num_inference_steps=1)
    torch.cuda.synchronize()
ms, fl = ops.Profiler.collect_list("igemm")
shapes, ops.Profiler.shapes = ops.Profiler.shapes, None
assert len(ms) == len(shapes), (len(ms), len(shapes))
agg = collections.OrderedDict()
for s, m, f in zip(shapes, ms, fl):
    e = agg.setdefault(s, [0, 0.0, 0.0]); e[0] += 1; e[1] += m; e[2] += f
tot = sum(ms)
print(f"# igemm launches of one denoise iteration, workload {a.workload}: {len(ms)} launches, {tot:.2f} ms, "
      f"{sum(fl) / tot / 1e9:.1f} TFLOP/s")
print(f"{'M':>8} {'N':>6} {'K':>6} k s u {'C1':>5} a e {'n':>4} {'ms':>9} {'%':>6} {'TFLOP/s':>8}")
for s, (n, m, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, kh, kw, st, up, c1, act, epi = s
    print(f"{M:8d} {N:6d} {K:6d} {kh}x{kw} {st} {up} {c1:5d} {act} {epi} {n:4d} {m:9.3f} {100 * m / tot:6.2f} {f / m / 1e9:8.1f}")
