"""Trajectory tracks -> ControlNet condition maps: the on-disk format (``{track id: [[x, y], ...]}`` JSON, e.g.
``/root/reference/dataset/VIPSeg/output_cotracker_all/*.json``), the scaling to the working resolution and the 13 + 1 maps the
reference draws with OpenCV before every ``pipeline(...)`` call (``scripts/run_inference_vipseg_json_repro.py:420-447``; the
training-time twin ``utils/dataset.py:741-766`` with its per-track channel flip).  The integer arithmetic runs on the host
exactly as in the reference (Python ``int()`` of the same float expressions); the drawing runs in ``pt_rasterize_tracks`` and
produces the ``[-1, 1]`` tensor ``controlnet_condition`` directly - no cv2, no PIL round trip.
"""
from __future__ import annotations

import json
from typing import Dict, List, Sequence

import torch

from . import hip, ops


def load_tracks(path: str) -> Dict[str, List[List[float]]]:
    """The reference's trajectory file: a JSON object ``{id: [[x, y], ...]}``; key order is draw order."""
    with open(path, "r") as f:
        d = json.load(f)
    if not isinstance(d, dict) or not all(isinstance(v, list) for v in d.values()):
        raise ValueError(f"{path}: expected a JSON object of point lists")
    return d


def scale_tracks(tracks: Dict[str, Sequence[Sequence[float]]], size: Sequence[int], original_size: Sequence[int],
                 mode: str = "inference") -> List[List[List[int]]]:
    """``size`` = [height, width] of the maps, ``original_size`` = (height, width[, 3]) of the frames the tracks were measured on.
    ``mode="inference"``: ``int(x * (W / W0))`` (``scripts/run_inference_vipseg_json_repro.py:431``); ``"dataset"``:
    ``int(x / W0 * W)`` (``utils/dataset.py:750``) - the two round differently for some sizes."""
    if mode not in ("inference", "dataset"):
        raise ValueError(f"mode {mode!r}: 'inference' or 'dataset'")
    out = []
    for key in tracks:
        if mode == "inference":
            out.append([[int(p[0] * (size[1] / original_size[1])), int(p[1] * (size[0] / original_size[0]))] for p in tracks[key]])
        else:
            out.append([[int(p[0] / original_size[1] * size[1]), int(p[1] / original_size[0] * size[0])] for p in tracks[key]])
    return out


def draw_list(scaled, start: int, end: int, mode: str = "inference"):
    """What the reference asks OpenCV for, per map ``t`` in [start, end): ``(kind, x0, y0, x1, y1, b, g, r, w)`` with kind 0 =
    ``cv2.line`` point t -> t+1, thickness w = 3, colour BGR (0,0,255); 1 = ``cv2.circle`` at point t+1, radius w = 3 filled,
    (0,255,0); 2 = ``cv2.cvtColor(BGR2RGB)`` - after every track in "dataset" mode (``utils/dataset.py:762``), once per map in
    "inference" mode.  The rasteriser consumes the scaled tracks directly; this list is its specification and what the tests
    compare with the reference's logged calls."""
    maps = []
    for t in range(start, end):
        calls = []
        for tr in scaled:
            calls.append((0, tr[t][0], tr[t][1], tr[t + 1][0], tr[t + 1][1], 0, 0, 255, 3))
            calls.append((1, tr[t + 1][0], tr[t + 1][1], 0, 0, 0, 255, 0, 3))
            if mode == "dataset":
                calls.append((2, 0, 0, 0, 0, 0, 0, 0, 0))
        if mode == "inference":
            calls.append((2, 0, 0, 0, 0, 0, 0, 0, 0))
        maps.append(calls)
    return maps


def trajectory_maps(tracks, size, original_size, num_frames: int = 14, mode: str = "inference", start: int = 0,
                    device="cuda", dtype=torch.float16) -> torch.Tensor:
    """``[num_frames, 3, H, W]`` in [-1, 1] on the device: ``num_frames - 1`` drawn maps (segment t -> t+1 as a red 3-px line,
    a green radius-3 disc at t+1) and the black last one (``scripts/...:446-447``), i.e. what
    ``image_processor.preprocess(validation_control_images)`` yields in the reference - pass it as ``controlnet_condition``."""
    scaled = scale_tracks(tracks, size, original_size, mode) if isinstance(tracks, dict) else tracks
    n_tracks = len(scaled)
    n_points = min((len(t) for t in scaled), default=2)
    n_maps = num_frames - 1
    if n_tracks and start + n_maps + 1 > n_points:
        raise ValueError(f"{n_maps} maps from step {start} need {start + n_maps + 1} points per track; the shortest has {n_points}")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("posetraj_amd.trajectory_maps: the rasteriser runs on the ROCm device (no CPU path exists)")
    if dtype not in (torch.float16, torch.float32):
        raise ValueError("dtype must be fp16 or fp32")
    if n_tracks:
        pts = torch.tensor([t[:n_points] for t in scaled], dtype=torch.int32).reshape(n_tracks, n_points, 2).to(device)
    else:                                                    # no tracks: every map is black; the kernel still wants a valid pointer
        pts, n_points, n_maps = torch.zeros((1, 2, 2), dtype=torch.int32, device=device), 2, 0
    H, W = int(size[0]), int(size[1])
    out = torch.empty((num_frames, 3, H, W), dtype=dtype, device=device)
    hip.check(hip.lib().pt_rasterize_tracks(pts.data_ptr(), n_tracks, n_points, start if n_tracks else 0, n_maps, num_frames, H, W,
                                            1 if mode == "dataset" else 0, 1 if dtype == torch.float32 else 0, out.data_ptr(),
                                            ops._stream()), "pt_rasterize_tracks")
    return out
