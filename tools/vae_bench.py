#!/usr/bin/env python3
"""Decode / encode time of the SVD-width AutoencoderKLTemporalDecoder on one MI355X, with the per-shape table of its
implicit-GEMM launches.
    python tools/vae_bench.py [--workload L|M|S] [--frames 14] [--chunk 8] > profiles/r04/vae_decode_<tag>.txt"""
import argparse, collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from posetraj_amd import AutoencoderKLTemporalDecoder, StableVideoDiffusionPipelineControlNet, ops
from posetraj_amd.pipeline_stable_video_diffusion_controlnet import tensor2vid

VAE = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512),
           layers_per_block=2, latent_channels=4, sample_size=768, scaling_factor=0.18215, force_upcast=True)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="L"); ap.add_argument("--frames", type=int, default=14); ap.add_argument("--chunk", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
H, W = bench.WORKLOADS[a.workload]
dev = torch.device("cuda:0")
vae = AutoencoderKLTemporalDecoder(**VAE).init_random_(seed=300, device=dev)
pipe = StableVideoDiffusionPipelineControlNet(vae=vae)
g = torch.Generator().manual_seed(1)
lat = (torch.randn(1, a.frames, 4, H // 8, W // 8, generator=g) * 0.18215).to(dev)


def decode():
    fr = pipe.decode_latents(lat, a.frames, a.chunk)
    return tensor2vid(fr, None, "pt")


out = decode(); torch.cuda.synchronize()
print(f"# AutoencoderKLTemporalDecoder (SVD widths, random init), {a.frames} x {H} x {W}, decode_chunk_size {a.chunk}; "
      f"finite={bool(torch.isfinite(out[0]).all())}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
ts = []
for _ in range(a.reps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); decode(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"decode_latents + tensor2vid('pt'): {min(ts) * 1e3:.1f} ms (best of {a.reps}; all {[round(t * 1e3, 1) for t in ts]}) "
      f"= {a.frames / min(ts):.1f} frames/s")
img = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(dev)
vae.encode(img); torch.cuda.synchronize(); t0 = time.perf_counter(); vae.encode(img); torch.cuda.synchronize()
print(f"encode (1 frame): {(time.perf_counter() - t0) * 1e3:.1f} ms")
ops.Profiler.shapes = []
with ops.Profiler():
    decode(); torch.cuda.synchronize()
ms, fl = ops.Profiler.collect_list("igemm")
shapes, ops.Profiler.shapes = ops.Profiler.shapes, None
agg = collections.OrderedDict()
for s, m, f in zip(shapes, ms, fl):
    e = agg.setdefault(s, [0, 0.0, 0.0]); e[0] += 1; e[1] += m; e[2] += f
tot = sum(ms)
print(f"# igemm launches of one decode: {len(ms)} launches, {tot:.2f} ms, {sum(fl) / 1e12:.1f} TFLOP, {sum(fl) / tot / 1e9:.1f} TFLOP/s")
print(f"{'M':>8} {'N':>6} {'K':>6} k s u {'C1':>5} a e {'n':>4} {'ms':>9} {'%':>6} {'TFLOP/s':>8}")
for s, (n, m, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, kh, kw, st, up, c1, act, epi = s
    print(f"{M:8d} {N:6d} {K:6d} {kh}x{kw} {st} {up} {c1:5d} {act} {epi} {n:4d} {m:9.3f} {100 * m / tot:6.2f} {f / m / 1e9:8.1f}")
