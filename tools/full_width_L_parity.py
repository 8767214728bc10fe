#!/usr/bin/env python3
"""BASELINE configs[2] end to end against the oracle: the full-width ControlNet + U-Net (1.52 B + 0.68 B parameters, seeded
random init), 14 x 576 x 1024 (latent 72 x 128), CFG, ONE loop iteration of the pipeline (hipGraph + two streams, as bench.py
runs it) on the MI355X vs the fp32 CPU oracle (123 TFLOP on the host: ~3 min on 16 cores, 22 GB).  Too slow for the -m gpu
suite, which runs the same check at configs[1]'s 40 x 72 latent (test_config1_full_width_loop_iteration_at_320x576);
output committed as profiles/r02/full_width_L_parity.txt.      python tools/full_width_L_parity.py"""
import sys, os, time, resource
sys.path.insert(0, os.getcwd())
import torch
from tests import parity as P
t = time.time()
cn_o, unet_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, "cuda:0", cfg=P.SVD_CFG, ce=P.SVD_CE)
print(f"build {time.time() - t:.1f} s", flush=True)
t = time.time()
r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(72, 128), device="cuda:0", nets=(cn_o, unet_o, cn_h, unet_h), seed=13,
                               use_graph=True, overlap_streams=True)
print(f"BASELINE configs[2], full-width networks, one CFG loop iteration at the 72 x 128 latent: rel-L2 vs fp32 oracle = {r:.3e}  "
      f"({time.time() - t:.1f} s, host peak {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB)", flush=True)
