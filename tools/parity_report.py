#!/usr/bin/env python3
"""Parity ladder (diagnostic; run on the MI355X): rel-L2 of the HIP ControlNet / U-Net / loop iteration against the CPU
oracle at three storage precisions (oracle/quant.py), per-tap errors at growing geometries, the full-width level-0
layer pair, and the scheduler goldens.  Test infrastructure: the oracle is only the checker here."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity as P  # noqa: E402

dev = "cuda:0"
fmt = lambda d: "  ".join(f"{k} {v:.2e}" for k, v in d.items())
print("== ladder, one network forward, tiny nets (pairs are a|b = ||a-b|| / ||b||)")
for hw in [(16, 16), (40, 72)]:
    d = P.net_ladder(dev, latent_hw=hw)
    for net, v in d.items():
        print(hw, net, fmt(v), flush=True)
print("== ladder, loop iterations (CFG + Euler), tiny nets")
for steps, hw in [(1, (16, 16)), (2, (16, 16)), (1, (40, 72))]:
    r, _, _, d = P.run_tiny_pipeline_parity(steps=steps, latent_hw=hw, device=dev, return_all=True,
                                            modes=("fp32", "fp16-fused", "fp16"))
    print(f"steps={steps} latent={hw}:", fmt(d), flush=True)
print("== per-tap, HIP vs fp32 oracle")
for hw in [(16, 16), (24, 40), (40, 72)]:
    cn_o, unet_o = P.build_oracle_nets(seed=0)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    i = P.tiny_inputs(seed=1, h=hw[0], w=hw[1])
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)
    j = {k: v.to(dev) for k, v in i.items()}
    down_h, mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), return_dict=False)
    print(hw, "cn taps:", " ".join(f"{P.rel_l2(a, b):.1e}" for a, b in zip(down_h, down_o)), "mid", f"{P.rel_l2(mid_h, mid_o):.1e}", flush=True)
    with torch.no_grad():
        y_o = unet_o(i["sample"], i["t"], i["ehs"], down_o, mid_o, return_dict=False, added_time_ids=i["ids"])[0]
    y_h = unet_h(j["sample"].half(), j["t"], j["ehs"].half(), [d.half().to(dev) for d in down_o], mid_o.half().to(dev), return_dict=False, added_time_ids=j["ids"])[0]
    print(hw, "unet:", f"{P.rel_l2(y_h, y_o):.2e}", flush=True)
if "--full" in sys.argv:
    t0 = time.time()
    for mode in ("fp32", "fp16-fused"):
        a, b = P.full_width_level0_block(dev, mode=mode)
        print(f"== full-width level-0 layer pair at 14x72x128 vs oracle[{mode}]: resblock {a:.2e}  transformer {b:.2e}  ({time.time() - t0:.0f} s)", flush=True)
# scheduler golden
import numpy as np  # noqa: E402
from posetraj_amd import EulerDiscreteScheduler, SVD_SCHEDULER_CONFIG  # noqa: E402
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sched.npz"))
for n in (2, 25):
    k = f"svd_n{n}_"
    s = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG); s.set_timesteps(n, device=dev)
    x = torch.from_numpy(g[k + "x0_f32"]).to(dev)
    for i in range(2):
        t = s.timesteps[i]
        xin = s.scale_model_input(x, t)
        print(n, i, "scale maxrel", float(np.max(np.abs(xin.cpu().numpy() - g[k + f"scaled{i}_f32"]) / (np.abs(g[k + f"scaled{i}_f32"]) + 1e-30))))
        mo = torch.from_numpy(g[k + f"model_out{i}_f32"]).to(dev)
        x = s.step(mo, t, x).prev_sample
        ref = g[k + f"prev{i}_f32"]
        print(n, i, "step rel", float(np.linalg.norm(x.cpu().numpy() - ref) / np.linalg.norm(ref)), "maxabs", float(np.abs(x.cpu().numpy() - ref).max()))
        x = torch.from_numpy(ref).to(dev)
