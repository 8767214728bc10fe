#!/usr/bin/env python3
"""The headline parity claim at a BASELINE geometry: the full-width ControlNet + U-Net (1.52 B + 0.68 B
parameters, seeded random init), BASELINE configs[1] = 14 x 320 x 576 (latent 40 x 72), CFG, the WHOLE 25-step Euler/Karras loop
of the pipeline (hipGraph + two streams, as bench.py and __call__ run it) on the MI355X against the fp32 CPU oracle
(25 x 33 TFLOP on the host: ~25 x 50 s).  A tool, not a test; output committed under profiles/r04/.
    python tools/full_width_M_25step_parity.py [--steps 25] [--latent 40 72]"""
import argparse, os, sys, time, resource
sys.path.insert(0, os.getcwd())
import torch
from tests import parity as P
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=25); ap.add_argument("--latent", type=int, nargs=2, default=(40, 72))
a = ap.parse_args()
t = time.time()
cn_o, unet_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, "cuda:0", cfg=P.SVD_CFG, ce=P.SVD_CE)
print(f"build {time.time() - t:.1f} s", flush=True)
t = time.time()
r = P.run_tiny_pipeline_parity(steps=a.steps, latent_hw=tuple(a.latent), device="cuda:0", nets=(cn_o, unet_o, cn_h, unet_h), seed=13,
                               use_graph=True, overlap_streams=True)
print(f"full-width networks, {a.steps}-step CFG loop at the {a.latent[0]} x {a.latent[1]} latent (14 x {a.latent[0] * 8} x {a.latent[1] * 8}): "
      f"rel-L2 of the final latents vs the fp32 oracle = {r:.3e}  ({time.time() - t:.1f} s, host peak "
      f"{resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB)", flush=True)
