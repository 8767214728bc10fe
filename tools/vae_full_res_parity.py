#!/usr/bin/env python3
"""The VAE at the benched geometry against the oracle (a tool, not a test: ~100 TFLOP on the host): AutoencoderKLTemporalDecoder at
the SVD widths (97.7 M parameters, seeded), F frames of 576 x 1024 (latent 72 x 128) decoded as ONE chunk - the largest
tensors the path sees (8.26 M pixel rows, temporal convolutions over an image 589 824 columns wide) - and one frame encoded,
HIP vs the fp32 CPU oracle.     python tools/vae_full_res_parity.py [--frames 14]"""
import argparse, os, sys, time, resource
sys.path.insert(0, os.getcwd())
import torch
from tests import parity as P                     # sets the host thread count
from tests.test_vae_gpu import _vaes, rel
from oracle import vae as OV
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=14); ap.add_argument("--latent", type=int, nargs=2, default=(72, 128))
ap.add_argument("--encode-only", action="store_true", help="only the encode of one frame: fp16 kernels and the fp32 path of force_upcast")
a = ap.parse_args()
dev = torch.device("cuda:0")
o, h = _vaes(dev, cfg=OV.svd_vae_config(), seed=77)
g = torch.Generator().manual_seed(2)
z = (torch.randn(a.frames, 4, *a.latent, generator=g) * 1.2).half().float()
if a.encode_only:
    x = (torch.rand(1, 3, a.latent[0] * 8, a.latent[1] * 8, generator=g) * 2 - 1) + 0.02 * torch.randn(1, 3, a.latent[0] * 8, a.latent[1] * 8, generator=g)
    t = time.time()
    with torch.no_grad():
        mref = o.encode(x).latent_dist.mode()
    t_or = time.time() - t
    r16 = rel(h.encode(x.to(dev)).latent_dist.mode(), mref)
    h.to(dtype=torch.float32)
    m32 = h.encode(x.to(dev)).latent_dist.mode(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.time(); h.encode(x.to(dev)); torch.cuda.synchronize(); ts.append(time.time() - t)
    h.to(dtype=torch.float16)
    t = time.time(); h.encode(x.to(dev)); torch.cuda.synchronize(); t16 = time.time() - t
    print(f"encode of one {a.latent[0] * 8} x {a.latent[1] * 8} frame (SVD widths), latent_dist.mode() vs the fp32 oracle ({t_or:.0f} s on the host): "
          f"fp32 path (vae.to(torch.float32), force_upcast) rel-L2 {rel(m32, mref):.3e} in {1000 * min(ts):.1f} ms;  fp16 kernels {r16:.3e} in {1000 * t16:.1f} ms", flush=True)
    sys.exit(0)
t = time.time()
got = h.decode(z.to(dev), num_frames=a.frames).sample
torch.cuda.synchronize()
print(f"HIP decode of {a.frames} x {a.latent[0] * 8} x {a.latent[1] * 8}: {time.time() - t:.2f} s (first call), finite={bool(torch.isfinite(got).all())}", flush=True)
t = time.time()
with torch.no_grad():
    ref = o.decode(z, num_frames=a.frames).sample
print(f"oracle decode: {time.time() - t:.0f} s; rel-L2 HIP vs fp32 oracle = {rel(got, ref):.3e}  "
      f"(max |diff| {float((got.cpu() - ref).abs().max()):.3e} on frames in [{float(ref.min()):.2f}, {float(ref.max()):.2f}])", flush=True)
x = (torch.rand(1, 3, a.latent[0] * 8, a.latent[1] * 8, generator=g) * 2 - 1).half().float()
with torch.no_grad():
    mref = o.encode(x).latent_dist.mode()
print(f"encode of one {a.latent[0] * 8} x {a.latent[1] * 8} frame: rel-L2 of latent_dist.mode() = {rel(h.encode(x.to(dev)).latent_dist.mode(), mref):.3e}; "
      f"host peak {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB", flush=True)
