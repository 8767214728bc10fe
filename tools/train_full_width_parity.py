#!/usr/bin/env python3
"""Gradient parity of the training step at the FULL model widths (U-Net 1.52 B frozen + ControlNet 0.68 B trainable, seeded random
init): ControlNetTrainer.loss_and_grads on the MI355X against fp32 torch autograd over the oracle's modules on the host
(oracle.train.training_step_grads - the restatement pinned to the reference script's own backward by tests/golden/train_grads.npz).
A tool, not a test (8.8 GB of fp32 weights and their autograd graph on the host); output committed under profiles/r04/.
    python tools/train_full_width_parity.py [--frames 2] [--latent 16 16]"""
import argparse, math, os, resource, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tests import parity as P
from oracle import train as OT

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--latent", type=int, nargs=2, default=(16, 16))
a = ap.parse_args()
t0 = time.time()
cn_o, un_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
with torch.no_grad():                                              # a ControlNet whose zero-convs have left zero: every gradient is live
    g = torch.Generator().manual_seed(11)
    for k, p in cn_o.named_parameters():
        if k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
            p.copy_((torch.randn(p.shape, generator=g) * 0.02).half().float())
from posetraj_amd import UNetSpatioTemporalConditionControlNetModel
from posetraj_amd.training import ControlNetTrainer
dev = torch.device("cuda:0")
un = UNetSpatioTemporalConditionControlNetModel(**P.SVD_CFG).load_state_dict(un_o.state_dict(), dev, keep_source=True)
cfg = dict(P.SVD_CFG, conditioning_embedding_out_channels=P.SVD_CE, down_block_types=un.config.down_block_types)
tr = ControlNetTrainer(cfg, cn_o.state_dict(), un, conditioning_dropout_prob=0.1, loss_scale=4096.0)
print(f"build {time.time() - t0:.1f} s", flush=True)
F, (h, w) = a.frames, a.latent
g = torch.Generator().manual_seed(12)
lat = (torch.randn(1, F, 4, h, w, generator=g) * 0.18215 * 5).half().float()
emb = torch.randn(1, 1, P.SVD_CFG["cross_attention_dim"], generator=g).half().float()
traj = (torch.rand(1, F, 3, h * 8, w * 8, generator=g) * 2 - 1).half().float()
noise = torch.randn(lat.shape, generator=g)
sig, rp, ran = torch.tensor([1.3]), torch.tensor([0.7]), F - 1
t1 = time.time()
r = tr.loss_and_grads(lat, emb, torch.tensor([127.0]), traj, noise=noise, sigmas=sig, random_p=rp, ran_idx=ran)
torch.cuda.synchronize()
print(f"MI355X step (forward + backward): {time.time() - t1:.2f} s; loss {r['loss']:.6f} (spatial {r['loss_spatial']:.6f})", flush=True)
got = {k: v.float().cpu() for k, v in tr.gradients().items()}
t2 = time.time()
ro = OT.training_step_grads(cn_o, un_o, lat, noise, sig, emb, torch.tensor([127.0]), traj, 0.18215, random_p=rp, conditioning_dropout_prob=0.1, ran_idx=ran)
print(f"host autograd: {time.time() - t2:.1f} s (peak {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB); loss {float(ro['loss']):.6f} "
      f"(spatial {float(ro['loss_spatial']):.6f})", flush=True)
want = ro["grads"]
names = sorted(want)
gw = torch.cat([want[k].reshape(-1).double() for k in names])
gg = torch.cat([got[k].reshape(-1).double() for k in names])
total = float((gg - gw).norm() / gw.norm())
big = float(gw.norm()) / math.sqrt(len(names))
rels = sorted(((float((got[k].double() - want[k].double()).norm() / want[k].double().norm()), k) for k in names if float(want[k].norm()) >= 0.05 * big), reverse=True)
dead = [k for k in names if float(want[k].abs().max()) == 0.0]
print(f"full-width training step, {F} frames at the {h} x {w} latent: loss rel. deviation {abs(r['loss'] / float(ro['loss']) - 1):.1e}; "
      f"all {len(names)} parameter gradients ({gw.numel() / 1e6:.0f} M values) rel-L2 {total:.3e}; sizeable tensors: median "
      f"{rels[len(rels) // 2][0]:.2e}, worst {rels[0][0]:.2e} ({rels[0][1]}); {len(dead)} tensors exactly zero on both sides: "
      f"{all(float(got[k].abs().max()) == 0.0 for k in dead)}")
