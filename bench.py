#!/usr/bin/env python3
"""Benchmark of the PoseTraj denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one whole clip: the 25-iteration denoise loop of
``pipeline_stable_video_diffusion_controlnet.py:481-583`` (ControlNetSDV + U-Net + CFG + Euler per iteration, plus the
once-per-clip condition-encoder pass) on one synthetic 14 x 576 x 1024 clip per GPU, fp16, random-init weights at full
SVD dimensions (1.52 B + 0.68 B parameters), inputs resident in HBM.  metric = denoised frames / s = clips * 14 / time.
Clips are independent, so N GPUs each run their own clip (weak scaling); the only collective is the start-up RCCL
broadcast of the packed weights from rank 0.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (fresh child
processes, before anything touches a GPU) - the same layout torch.distributed.run would give.

Extra legs on rank 0:
  roofline      the dominant kernel family - the implicit-GEMM convolutions / linear layers (igemm*_kernel) and the two fused forms that
                took launches over from it (ffn320_kernel: GEGLU feed-forward, lnlin320_kernel: LayerNorm + Q|K|V projection):
                algorithmic flops / hipEvent time of every launch of one extra clip, against the 2.5 PFLOP/s dense fp16 MFMA peak.
                Also reported: the spatial attention kernel and the whole path (executed flops / clip wall time).
  cpu_baseline  the CPU oracle (oracle/, fp32 PyTorch on the host cores): ONE real loop iteration of the full-width U-Net +
                ControlNet at the benched geometry itself (14 x 576 x 1024: latent 72 x 128, ~123 TFLOP of fp32, ~200 s on 15 threads)
                after a 64 x 64 px warm-up; reference-executed flops are counted exactly with meta tensors.  It runs in a child process
                started AFTER the timed clips (beside the roofline clip, whose numbers are device-side event brackets).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def traffic_per_launch(args):
    """HBM-side bytes per igemm launch for this workload, from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes over tools/traffic_run.py, gfx950 correction applied: tools/traffic_summary.py).
    bench.py cannot collect PMC counters itself (they need rocprofv3 passes of their own), so this is a REPLAY of a committed
    measurement - valid only for the build it was taken on: the summary records the sha256 of the kernel sources
    (posetraj_amd.hip.source_digest) and a summary of another build is refused (None), as is a workload with no summary."""
    if args.camera:
        return None
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    best = None
    for dirpath, _, files in os.walk(root):
        for f in sorted(files):
            if f.startswith(f"summary_{args.workload}_") and f.endswith(".json"):
                cand = os.path.join(dirpath, f)              # latest round directory (profiles/rNN/...), then latest tag
                best = cand if best is None or cand > best else best
    if best is None:
        return None
    with open(best) as fh:
        d = json.load(fh)
    from posetraj_amd import hip
    if d.get("csrc_sha256") != hip.source_digest():          # kernels changed since the counters were read: stale
        return None
    return {"hbm_bytes_per_launch": round(d["hbm_bytes_per_launch"]), "algorithmic_bytes_per_launch": round(d["algorithmic_bytes_per_launch"]),
            "source": os.path.relpath(best, os.path.dirname(os.path.abspath(__file__)))}


PEAK_FP16_DENSE_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
SVD = dict(block_out_channels=(320, 640, 1280, 1280), num_attention_heads=(5, 10, 20, 20), cross_attention_dim=1024,
           addition_time_embed_dim=256, projection_class_embeddings_input_dim=768, layers_per_block=2, num_frames=14)
SVD_VAE = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512),
               layers_per_block=2, latent_channels=4, sample_size=768, scaling_factor=0.18215, force_upcast=True)
CLIP_VIT_H = dict(hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=32, num_attention_heads=16,
                  num_channels=3, image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-5)
WORKLOADS = {"L": (576, 1024), "M": (320, 576), "S": (128, 128)}


def synth_tracks(frames, H, W, seed, n_tracks=8):
    """8 smooth random tracks of `frames` points in the reference's on-disk format {id: [[x, y], ...]} (dataset/VIPSeg/
    output_cotracker_all/*.json), measured on an H x W frame."""
    rng = np.random.default_rng(seed)
    tracks = {}
    for i in range(n_tracks):
        p = rng.uniform([0.15 * W, 0.15 * H], [0.85 * W, 0.85 * H])
        v = rng.normal(0, 0.02 * W, size=2)
        pts = []
        for _f in range(frames):
            pts.append([int(p[0]), int(p[1])])
            v = 0.8 * v + rng.normal(0, 0.01 * W, size=2)
            p = np.clip(p + v, 3, [W - 4, H - 4])
        tracks[str(i)] = pts
    return tracks


def synth_control_maps(frames, H, W, seed, device):
    """The control maps of scripts/run_inference_vipseg_json_repro.py:426-449 for synthetic tracks: frames - 1 maps with a red
    3-px segment + a green radius-3 disc per track and step, last map black, in [-1, 1] - drawn by the package's own
    rasteriser (posetraj_amd.trajectory -> pt_rasterize_tracks), the way a user without cv2 would build them."""
    from posetraj_amd.trajectory import trajectory_maps
    return trajectory_maps(synth_tracks(frames, H, W, seed), [H, W], (H, W, 3), num_frames=frames, device=device)


def synth_clip(height, width, frames, xdim, seed, device, init_noise_sigma):
    g = torch.Generator().manual_seed(seed)
    h, w = height // 8, width // 8
    latents = torch.randn(1, frames, 4, h, w, generator=g) * float(init_noise_sigma)
    mode = torch.randn(1, 4, h, w, generator=g)
    image_latents = torch.cat([torch.zeros_like(mode), mode])
    e = torch.randn(1, 1, xdim, generator=g)
    emb = torch.cat([torch.zeros_like(e), e])
    cond = synth_control_maps(frames, height, width, seed, device).unsqueeze(0)
    cond = torch.cat([cond] * 2)
    return (latents.to(device), image_latents.to(device, torch.float16), emb.to(device, torch.float16),
            cond.to(device, torch.float16))


def shard(items, rank, world):
    """Independent units (clips) are dealt round-robin; no data-path collective exists."""
    return list(items)[rank::world]


def packed_tensors(obj, seen=None, out=None, cuda_only=True):
    """Every device tensor reachable from a model (packed weights, norm vectors), each once."""
    seen = set() if seen is None else seen
    out = [] if out is None else out
    if id(obj) in seen:
        return out
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda or not cuda_only:
            out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            packed_tensors(o, seen, out, cuda_only)
    elif isinstance(obj, dict):
        for o in obj.values():
            packed_tensors(o, seen, out, cuda_only)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("posetraj_amd"):
        for o in vars(obj).values():
            packed_tensors(o, seen, out, cuda_only)
    return out


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs behind a 16-CPU quota; 256 threads on 16 CPUs do not finish)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


class PowerSampler:
    """Socket power of the card under test while the timed clips run (hwmon sysfs, every 100 ms, a daemon thread on rank 0
    at N = 1).  The card is found by its PCI address (torch device properties -> /sys/class/drm/card*/device); when
    that fails, the card with the highest median power.  Context for the roofline fraction - under matrix-core load the
    MI355X sits near its power cap and well below the 2.4 GHz the dense peak is quoted at; never
    fatal, absent from the line when the sensors are unreadable."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input") +
                            glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
        self.mine = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for f in self.files:
                if bdf in os.path.realpath(f.split("/hwmon/")[0]):
                    self.mine = f
        except Exception:
            pass
        if self.mine is not None:
            self.files = [self.mine]
        self.rows, self._stop = [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _read(self, f):
        try:
            return float(open(f).read()) / 1e6
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            self.rows.append([self._read(f) for f in self.files])
            self._stop.wait(0.1)

    def start(self):
        if self.files:
            self._t.start()
        return self

    def result(self):
        self._stop.set()
        try:
            if not self.files or len(self.rows) < 5:
                return None
            best = None
            for i, f in enumerate(self.files):
                v = sorted(r[i] for r in self.rows if r[i] is not None)
                if v and (best is None or v[len(v) // 2] > best[0]):
                    cap = self._read(f.rsplit("/", 1)[0] + "/power1_cap")
                    best = (v[len(v) // 2], {"median": round(v[len(v) // 2]), "mean": round(sum(v) / len(v), 1), "max": round(v[-1]), "cap": None if cap is None else round(cap),
                                             "samples": len(v), "card": "pci" if self.mine else "highest median of the node"})
            return best[1] if best else None
        except Exception:
            return None


class CpuBaselineChild:
    """The CPU leg in a child process (never touches the GPU), started after the timed clips and collected at the end, with
    a wall-clock budget so the bench line is always printed."""

    def __init__(self, frames, height, width, infer_steps, sample_latent, budget_s=600):
        import subprocess
        self.t0, self.budget = time.time(), budget_s
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--frames", str(frames),
               "--child-hw", str(height), str(width), "--infer-steps", str(infer_steps),
               "--child-sample-latent", str(sample_latent[0]), str(sample_latent[1])]
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        self.p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)

    def result(self):
        import subprocess
        fail = {"value": None, "unit": "frames/s", "cores": usable_cores(), "kind": "port"}
        try:
            out, err = self.p.communicate(timeout=max(1.0, self.budget - (time.time() - self.t0)))
        except subprocess.TimeoutExpired:
            self.p.kill()
            self.p.communicate()
            return dict(fail, sample=f"CPU leg exceeded its {self.budget} s budget and was stopped")
        for ln in reversed(out.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return dict(fail, sample=f"CPU leg failed: {err.strip()[-300:]}")


def cpu_baseline(frames=14, latent=(40, 72), warm_latent=(8, 8)):
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import nets as ON
    cores = max(1, usable_cores() - 1)             # one core stays with the process that drives the GPU
    torch.set_num_threads(cores)
    cfg = ON.svd_config()

    def build(device):
        with contextlib.redirect_stdout(io.StringIO()), torch.device("meta"):
            u = ON.UNetSpatioTemporalConditionControlNetModel(**cfg)
            c = ON.ControlNetSDVModel(**cfg)
        return u, c

    def flops(h, w):
        u, c = build("meta")
        with torch.device("meta"), FlopCounterMode(display=False) as fc:
            x, t, e = torch.empty(2, frames, 8, h, w), torch.empty(()), torch.empty(2, 1, 1024)
            ids, cond = torch.empty(2, 3), torch.empty(2, frames, 3, h * 8, w * 8)
            d, m = c(x, t, e, ids, controlnet_cond=cond, return_dict=False)
            u(x, t, e, d, m, return_dict=False, added_time_ids=ids)
        return fc.get_total_flops()

    u, c = build("meta")
    u, c = u.to_empty(device="cpu"), c.to_empty(device="cpu")
    base = torch.randn(1 << 20, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        for mdl in (u, c):
            for n, p in mdl.named_parameters():
                if "norm" in n and n.endswith("weight"):
                    p.fill_(1.0)
                elif n.endswith("bias"):
                    p.zero_()
                elif n.endswith("mix_factor"):
                    p.fill_(0.5)
                else:                                   # cheap pseudo-random fill, fan-in scaled (values only need to be sane)
                    k = p.numel()
                    p.view(-1).copy_(base.repeat((k + base.numel() - 1) // base.numel())[:k])
                    p.mul_((p[0].numel() if p.ndim > 1 else 1) ** -0.5)
    g = torch.Generator().manual_seed(1)
    e = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6, 128, 0.02]] * 2)
    t = torch.tensor(1.0)

    def one_iteration(h, w):
        x = torch.randn(2, frames, 8, h, w, generator=g)
        cond = torch.rand(2, frames, 3, h * 8, w * 8, generator=g) * 2 - 1
        t0 = time.perf_counter()
        with torch.no_grad():
            d, m = c(x, t, e, ids, controlnet_cond=cond, return_dict=False)
            u(x, t, e, d, m, return_dict=False, added_time_ids=ids)
        return time.perf_counter() - t0

    t_warm = one_iteration(*warm_latent)           # thread pools, allocator, weight pages
    t_step = one_iteration(*latent)                # the timed sample: one real iteration
    return dict(t_step=t_step, t_warm=t_warm, flops_sample=flops(*latent), cores=cores, flops_fn=flops)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh children (this process has not touched a GPU and never
    will), one per GPU, wired like torch.distributed.run would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", PT_BENCH_SELF_SPAWNED="1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def broadcast_packed(tensors, src=0, bucket_bytes=1 << 30):
    """Start-up weight broadcast: the packed tensors travel as a few flat buckets (<= 1 GiB each, per dtype) instead of
    one small collective per tensor.  Returns (GB moved, seconds, number of collectives)."""
    import torch.distributed as dist
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    gb, n_coll = 0.0, 0
    t0 = time.perf_counter()
    for dt, ts in by_dtype.items():
        i = 0
        while i < len(ts):
            j, nbytes = i, 0
            while j < len(ts) and (j == i or nbytes + ts[j].numel() * ts[j].element_size() <= bucket_bytes):
                nbytes += ts[j].numel() * ts[j].element_size()
                j += 1
            flat = torch.cat([t.reshape(-1) for t in ts[i:j]])
            dist.broadcast(flat, src=src)
            off = 0
            for t in ts[i:j]:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            gb += nbytes / 1e9
            n_coll += 1
            i = j
    if tensors and tensors[0].is_cuda:
        torch.cuda.synchronize()
    return gb, time.perf_counter() - t0, n_coll


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="L", choices=list(WORKLOADS))
    ap.add_argument("--infer-steps", type=int, default=25)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--camera", action="store_true",
                    help="BASELINE configs[4]: controlnet_sdv_cam (camera-disentangle branch) with per-frame R|T conditioning")
    ap.add_argument("--no-decode", action="store_true", help="skip the VAE-decode leg (reported beside the headline, never in it)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time the whole image-to-video call (resize + CLIP ViT-H + VAE encode + loop + VAE decode + tensor2vid) on "
                         "random-init full-size models; reported beside the headline")
    ap.add_argument("--train-step", action="store_true",
                    help="also time the reference's ControlNet training step (scripts/train_svd_traj_VIPSeg_14.py:1264-1425; start_ft.sh: 14 x 320 x "
                         "576, batch 1, fp16 mixed precision): forward + backward + AdamW on the same full-size networks; beside the headline")
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="throughput mode, reported as its own field beside the headline (which stays one clip per GPU): K independent clips "
                         "batched through every launch of the loop (SURVEY Appendix B: the small-M layers of levels 2-3 fill more of the chip)")
    ap.add_argument("--no-overlap", action="store_true", help="ControlNet and U-Net encoder on one stream (default: two)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--child-hw", type=int, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-sample-latent", type=int, nargs=2, default=(40, 72), help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="(test hook) start the ranks, rendezvous over gloo on the CPU, shard the clips, print the layout and exit")
    args = ap.parse_args()

    if args.cpu_baseline_child:                    # CPU-only child of the cpu_baseline leg: never touches the GPU
        height, width = args.child_hw
        sl = tuple(args.child_sample_latent)
        cb = cpu_baseline(frames=args.frames, latent=sl)
        f_full = cb["flops_fn"](height // 8, width // 8)
        t_full_step = cb["t_step"] * f_full / cb["flops_sample"]
        same = (height // 8, width // 8) == sl
        print(json.dumps({
            "value": round(args.frames / (args.infer_steps * t_full_step), 6), "unit": "frames/s", "cores": cb["cores"],
            "kind": "port",
            "sample": (f"oracle/ (fp32 PyTorch restatement of the reference path), full-width U-Net + ControlNet, "
                       f"{args.frames} frames at {sl[0] * 8}x{sl[1] * 8} px (latent {sl[0]}x{sl[1]}), CFG batch 2: ONE loop iteration = "
                       f"{cb['t_step']:.1f} s for {cb['flops_sample'] / 1e12:.2f} TFLOP on {cb['cores']} threads (after a 64x64 px warm-up "
                       f"iteration of {cb['t_warm']:.1f} s); " +
                       ("" if same else f"scaled to {height}x{width} by reference-executed flops ({f_full / 1e12:.2f} TFLOP/iteration), ") +
                       f"x {args.infer_steps} iterations per clip; extrapolated, baseline only")}))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    import torch.distributed as dist
    if args.rendezvous_only:                       # the launch / sharding path of the N > 1 bench without any GPU
        if world > 1:
            dist.init_process_group("gloo")
        mine = shard(list(range(world)), rank, world)
        got = torch.tensor([1234 + c for c in mine], dtype=torch.long)
        if world > 1:
            allv = [torch.zeros_like(got) for _ in range(world)]
            dist.all_gather(allv, got)
            got = torch.cat(allv)
        if rank == 0:
            print(json.dumps({"world": world, "clip_seeds": sorted(int(v) for v in got), "launcher": "self" if os.environ.get("PT_BENCH_SELF_SPAWNED") else "external"}))
        if world > 1:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from posetraj_amd import (ControlNetSDVModel, EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet,
                              SVD_SCHEDULER_CONFIG, UNetSpatioTemporalConditionControlNetModel, ops)
    height, width = WORKLOADS[args.workload]
    cpu_child = None
    unet = UNetSpatioTemporalConditionControlNetModel(**SVD).init_random_(seed=100 + rank, device=dev)
    cn = ControlNetSDVModel(**SVD, camera=args.camera).init_random_(seed=200 + rank, device=dev)
    bcast_gb, bcast_s, bcast_n = 0.0, 0.0, 0
    if world > 1:                                  # start-up broadcast of the packed weights over RCCL / xGMI
        bcast_gb, bcast_s, bcast_n = broadcast_packed(packed_tensors(unet) + packed_tensors(cn), src=0)
    sched = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet, controlnet=cn, scheduler=sched)
    sched.set_timesteps(args.infer_steps)
    clip_id = shard(list(range(world)), rank, world)[0]          # one independent clip per GPU, dealt rank::world
    clip = synth_clip(height, width, args.frames, SVD["cross_attention_dim"], 1234 + clip_id, dev, sched.init_noise_sigma)

    cam = None
    if args.camera:                                # small per-frame rotations about one axis + translation, frame 0 subtracted
        ang = torch.linspace(0, 0.3, args.frames)
        rt = torch.zeros(args.frames, 12)
        rt[:, 0], rt[:, 1], rt[:, 3], rt[:, 4], rt[:, 8] = torch.cos(ang) - 1, -torch.sin(ang), torch.sin(ang), torch.cos(ang) - 1, 0
        rt[:, 9] = torch.linspace(0, 0.5, args.frames)
        cam = torch.cat([rt.unsqueeze(0)] * 2).to(dev, torch.float16)

    def run_clip(use_graph=not args.no_graph):
        lat, il, emb, cond = clip
        if not use_graph:
            cn._cond_cache = None                  # the once-per-clip condition encoder is part of every clip
        # graph mode: the inputs are copied into the graph's static buffers, which re-runs the condition encoder too
        return pipe.denoise(lat, il, emb, cond, num_inference_steps=args.infer_steps, camera_cond=cam, use_graph=use_graph,
                            overlap_streams=use_graph and not args.no_overlap)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = run_clip()
    fence()
    power = PowerSampler(dev.index or 0).start() if (rank == 0 and world == 1) else None
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = run_clip()
    fence()
    elapsed = time.perf_counter() - t0
    power_w = power.result() if power is not None else None
    # the CPU leg starts only now: beside the timed clips its host threads and the launch thread perturbed each other (and the
    # power sampler).  It overlaps the roofline clip below, whose numbers are hipEvent brackets on the device.
    if world == 1 and not args.no_cpu_baseline:
        # one REAL iteration of the benched workload itself (72 x 128: ~123 TFLOP of fp32, ~200 s on 15 threads, inside
        # the child's budget; rounds 1-4 timed 40 x 72 and scaled by flops)
        sample_latent = (height // 8, width // 8)
        cpu_child = CpuBaselineChild(args.frames, height, width, args.infer_steps, sample_latent)
    # roofline leg: one more clip, outside the timed region, launched eagerly (a graph replay bypasses the C-ABI entry
    # points, so their hipEvent brackets would see nothing) with events around every igemm / attention launch on the
    # launch stream.  Same kernels, same shapes, same order as the timed clips.
    # decode leg (beside the headline, never in it): the step right after the loop - decode_latents at the
    # reference script's decode_chunk_size = 8 + tensor2vid - on a random-init AutoencoderKLTemporalDecoder at the SVD widths
    extra = {}
    if rank == 0 and not args.no_decode:
        from posetraj_amd import AutoencoderKLTemporalDecoder
        from posetraj_amd.pipeline_stable_video_diffusion_controlnet import tensor2vid
        vae = AutoencoderKLTemporalDecoder(**SVD_VAE).init_random_(seed=300, device=dev)
        dpipe = StableVideoDiffusionPipelineControlNet(vae=vae)
        lat_dec = (out.float() / max(float(out.float().std()), 1e-6) * 0.18215).to(dev)     # unit-variance latents, VAE-scaled

        def decode_once():
            return tensor2vid(dpipe.decode_latents(lat_dec, args.frames, 8), None, "pt")
        decode_once(); torch.cuda.synchronize()
        td = []
        for _ in range(3):
            t0d = time.perf_counter(); fr = decode_once(); torch.cuda.synchronize(); td.append(time.perf_counter() - t0d)
        extra["decode"] = {"ms_per_clip": round(1000 * min(td), 1), "decode_chunk_size": 8, "frames_per_s": round(args.frames / min(td), 1),
                           "vae": "AutoencoderKLTemporalDecoder at SVD widths (97.7 M params, random init)",
                           "share_of_loop_time": round(min(td) / (elapsed / args.steps), 4), "output_finite": bool(torch.isfinite(fr[0]).all().item())}
        if args.end_to_end:
            from posetraj_amd import CLIPVisionModelWithProjection
            clipm = CLIPVisionModelWithProjection(**CLIP_VIT_H).init_random_(seed=400, device=dev)
            epipe = StableVideoDiffusionPipelineControlNet(vae=vae, image_encoder=clipm, unet=unet, controlnet=cn, scheduler=sched)
            g = torch.Generator().manual_seed(7)
            image = torch.rand(1, 3, height, width, generator=g)
            maps = clip[3][0]                                            # the [F, 3, H, W] control maps in [-1, 1]

            def e2e():
                return epipe(image, maps, height=height, width=width, num_frames=args.frames, decode_chunk_size=8,
                             num_inference_steps=args.infer_steps, generator=torch.Generator().manual_seed(1), output_type="pt").frames
            e2e(); torch.cuda.synchronize()
            t0e = time.perf_counter(); fr = e2e(); torch.cuda.synchronize()
            te = time.perf_counter() - t0e
            extra["end_to_end"] = {"s_per_clip": round(te, 3), "frames_per_s": round(args.frames / te, 3),
                                   "stages": "resize 224 + CLIP ViT-H/14 (632 M) + VAE encode + loop + VAE decode (chunk 8) + tensor2vid('pt')",
                                   "output_finite": bool(torch.isfinite(fr[0]).all().item())}
            del clipm, epipe
        del vae, dpipe
        torch.cuda.empty_cache()
    if rank == 0 and args.train_step and not args.camera:
        # training leg (beside the headline, never in it): the other caller of the same modules - the reference's
        # fine-tuning geometry (start_ft.sh: --width=576 --height=320, 14 frames, batch 1, fp16 mixed precision), forward +
        # backward through the frozen U-Net's up path + AdamW over the ControlNet's 682 M parameters.  Measured in a process of
        # its own (tools/train_step_bench.py --json), like a training job runs: next to this process's captured graphs and
        # their streams the step's three streams gain nothing (profiles/r04/train_step_in_process_ab.txt)
        import subprocess
        tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "train_step_bench.py")
        r = subprocess.run([sys.executable, tool, "--json", "--steps", "7", "--warmup", "3", "--frames", str(args.frames)], capture_output=True, text=True, timeout=1200)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"bench.py --train-step: tools/train_step_bench.py failed ({r.returncode}): {r.stderr[-1500:]}")
        ts = json.loads(lines[-1])
        ts["workload"] = f"{args.frames}x320x576, batch 1, fp16 mixed precision (fp32 master weights), temporal + 0.5 spatial loss, AdamW; own process"
        ts["frac_of_mfma_peak"] = round(ts["matrix_TFLOP_per_step"] / (ts["ms_per_step"] * 1e-3) / PEAK_FP16_DENSE_TFLOPS, 4)
        extra["train_step"] = ts
    if rank == 0 and args.clips_per_gpu > 1 and not args.camera:
        # throughput mode (VERDICT r04 #1c; beside the headline, never in it): K independent clips through one loop - frame batch
        # 2 K x 14 per launch.  Same kernels, same per-clip arithmetic (GroupNorm statistics per sample, attention per frame / per
        # position, the CFG halves paired per clip); what changes is the tile count of the small-M launches and the launch tails.
        Kc = args.clips_per_gpu
        cl = [synth_clip(height, width, args.frames, SVD["cross_attention_dim"], 1234 + clip_id + 1000 * j, dev, sched.init_noise_sigma) for j in range(Kc)]
        latK = torch.cat([c[0] for c in cl])
        ilK = torch.cat([c[1][:1] for c in cl] + [c[1][1:] for c in cl])          # unconditional halves first
        embK = torch.cat([c[2][:1] for c in cl] + [c[2][1:] for c in cl])
        condK = torch.cat([c[3][:1] for c in cl] + [c[3][1:] for c in cl])
        pipeK = StableVideoDiffusionPipelineControlNet(unet=unet, controlnet=cn, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))

        def run_k():
            return pipeK.denoise(latK, ilK, embK, condK, num_inference_steps=args.infer_steps, use_graph=not args.no_graph,
                                 overlap_streams=(not args.no_graph) and not args.no_overlap)
        outK = run_k(); torch.cuda.synchronize()
        pK = PowerSampler(dev.index or 0).start()
        t0k = time.perf_counter()
        for _ in range(args.steps):
            outK = run_k()
        torch.cuda.synchronize()
        tk = (time.perf_counter() - t0k) / args.steps
        pwK = pK.result()
        extra["throughput_mode"] = {"clips_per_gpu": Kc, "ms_per_step": round(1000 * tk, 2), "frames_per_s": round(Kc * args.frames / tk, 4),
                                    "vs_one_clip_per_gpu": round((Kc * args.frames / tk) / (args.steps * args.frames / elapsed), 4),
                                    "output_finite": bool(torch.isfinite(outK).all().item()),
                                    "first_clip_equals_the_headline_clip_rel_l2": round(float((outK[0].float() - out[0].float()).norm() / out[0].float().norm()), 6),
                                    "socket_power_W": pwK}
        del pipeK, latK, ilK, embK, condK, outK
        torch.cuda.empty_cache()
    prof = {}
    if rank == 0 and not args.no_profile:
        with ops.Profiler():
            tc0 = time.perf_counter()
            run_clip(use_graph=False)
            torch.cuda.synchronize()
            prof["clip_s"] = time.perf_counter() - tc0
        prof["igemm"] = ops.Profiler.collect("igemm")
        prof["attn_spatial"] = ops.Profiler.collect("attn_spatial")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    finite = bool(torch.isfinite(out).all().item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = world * args.steps * args.frames / elapsed
    line = {
        "metric": "denoised frames/sec", "value": round(value, 4), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"SVD-img2vid U-Net + controlnet_sdv{'_cam (camera R|T)' if args.camera else ''}, {args.frames}x{height}x{width}, "
                               f"{args.infer_steps} Euler steps, CFG, 1 clip per GPU (BASELINE configs[{2 if args.workload == 'L' else 1}])",
                   "weights": "random-init at full SVD dimensions (1524.6 M + 682.0 M params)", "clips_per_gpu_per_step": 1,
                   "latent": [height // 8, width // 8], "output_finite": finite, "hipgraph": not args.no_graph, "two_streams": (not args.no_graph) and (not args.no_overlap),
                   "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
                   "weight_broadcast_GB": round(bcast_gb, 2), "weight_broadcast_collectives": bcast_n,
                   "weight_broadcast_GB/s": (round(bcast_gb / bcast_s, 1) if bcast_s > 0 else None)},
    }
    if prof:
        ig, at = prof["igemm"], prof["attn_spatial"]
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        line["roofline"] = {
            "bound": "mfma", "kernel": "igemm family: igemm*_kernel (implicit-GEMM conv / linear) + ffn320_kernel (fused GEGLU feed-forward) + lnlin320_kernel (LayerNorm + QKV), v_mfma_f32_16x16x32_f16",
            "achieved": round(ach, 1), "peak": PEAK_FP16_DENSE_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP16_DENSE_TFLOPS, 4), "traffic": traffic_per_launch(args),
            "launches": ig["launches"], "avg_launch_us": round(1000 * ig["ms"] / max(ig["launches"], 1), 2),
            "flops_per_launch_avg": round(ig["flops"] / max(ig["launches"], 1) / 1e9, 3),
            "attn_spatial": {"achieved": round(at["flops"] / (at["ms"] * 1e-3) / 1e12, 1) if at["ms"] > 0 else 0.0,
                             "launches": at["launches"], "ms": round(at["ms"], 2)},
            "path": {"executed_TFLOP_per_clip": round((ig["flops"] + at["flops"]) / 1e12, 2),
                     "clip_s_profiled": round(prof["clip_s"], 3),
                     "frac_of_mfma_peak": round((ig["flops"] + at["flops"]) / 1e12 / (elapsed / args.steps) / PEAK_FP16_DENSE_TFLOPS, 4),   # executed flops / TIMED clip
                     "igemm_share_of_clip_time": round(ig["ms"] * 1e-3 / prof["clip_s"], 3),
                     "attn_share_of_clip_time": round(at["ms"] * 1e-3 / prof["clip_s"], 3)},
        }
    if power_w is not None and "roofline" in line:
        line["roofline"]["socket_power_W_timed_region"] = power_w
        # the loop is power-limited: what a change buys is joules per clip.  Mean socket power over the timed
        # region x time per clip, and per executed matrix flop (igemm + spatial attention of the roofline clip)
        j_clip = power_w["mean"] * elapsed / args.steps
        line["roofline"]["energy"] = {"J_per_clip": round(j_clip, 1), "J_per_iteration": round(j_clip / args.infer_steps, 2),
                                      "pJ_per_flop": round(j_clip / max(ig["flops"] + at["flops"], 1.0) * 1e12, 3)}
    line.update(extra)
    if cpu_child is not None:
        line["cpu_baseline"] = cpu_child.result()
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
