#!/usr/bin/env python3
"""One denoise iteration of workload L/M/S with the igemm launch shapes recorded in launch order, for joining with a
rocprofv3 --pmc pass (FETCH_SIZE / WRITE_SIZE are per dispatch; the last len(shapes) igemm dispatches of the process
are this iteration's, in the same order).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python3 tools/traffic_run.py out/shapes.json
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from posetraj_amd import (ControlNetSDVModel, EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet,
                          SVD_SCHEDULER_CONFIG, UNetSpatioTemporalConditionControlNetModel, hip, ops)

out = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "L"
H, W = bench.WORKLOADS[workload]
dev = torch.device("cuda:0")
unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=100, device=dev)
cn = ControlNetSDVModel(**bench.SVD).init_random_(seed=200, device=dev)
sched = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
pipe = StableVideoDiffusionPipelineControlNet(unet=unet, controlnet=cn, scheduler=sched)
sched.set_timesteps(2)
clip = bench.synth_clip(H, W, 14, 1024, 1234, dev, sched.init_noise_sigma)
pipe.denoise(*clip, num_inference_steps=1)                                   # warm-up (and condition-encoder cache)
torch.cuda.synchronize()
ops.Profiler.shapes = []
with ops.Profiler():
    pipe.denoise(*clip, num_inference_steps=1)
    torch.cuda.synchronize()
shapes, ops.Profiler.shapes = ops.Profiler.shapes, None
with open(out, "w") as f:
    json.dump({"workload": workload, "csrc_sha256": hip.source_digest(),      # the build the counters of THIS run belong to
               "columns": ["M", "N", "K", "KH", "KW", "stride", "upsample2x", "C1", "act", "epi"],
               "shapes": [list(s) for s in shapes]}, f)
print(f"{len(shapes)} igemm launches recorded -> {out}")
