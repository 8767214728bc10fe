#!/usr/bin/env python3
"""Time of the trajectory-map rasteriser (pt_rasterize_tracks) at 14 x 576 x 1024 for 8 and 64 tracks; algorithmic bytes =
the [14, 3, H, W] fp16 tensor it writes (49.5 MB) - the kernel reads only the track table.   python tools/raster_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from posetraj_amd.trajectory import trajectory_maps
dev = torch.device("cuda:0")
H, W, F = 576, 1024, 14
for n in (8, 64, 512):
    tracks = bench.synth_tracks(F, H, W, 1, n_tracks=n)
    for _ in range(3):
        m = trajectory_maps(tracks, [H, W], (H, W, 3), num_frames=F, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m = trajectory_maps(tracks, [H, W], (H, W, 3), num_frames=F, device=dev); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(f"{n:4d} tracks: {best * 1e3:7.1f} us (host scaling + table upload + kernel), {m.numel() * 2 / best / 1e6:6.1f} GB/s of output written, "
          f"{int((m[:-1] > -1).any(dim=1).sum())} pixels drawn")
