"""Oracle restatement of the trajectory-map construction that feeds ``controlnet_condition``:

* inference: ``/root/reference/scripts/run_inference_vipseg_json_repro.py:426-449`` - tracks ``{id: [[x, y], ...]}`` scaled with
  ``int(x * (W / W0))``, 13 maps (segment t -> t+1 as a red 3-px line + a green radius-3 disc at t+1, drawn in BGR and flipped
  to RGB once per map), a black 14th map;
* training: ``/root/reference/utils/dataset.py:741-766`` (``draw_traj``) - scaled with ``int(x / W0 * W)`` and with the
  ``cvtColor`` INSIDE the per-track loop (``:762``): the channel order flips after every track, so tracks alternate colours.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Pinning status
  PINNED (reference run, ``tests/golden/tracks.npz``): the integer arithmetic and the draw list - ``make_golden.py: gen_tracks``
  executes the reference's own statements (extracted from the two files at generation time) against a RECORDING ``cv2`` stand-in
  that logs every ``line`` / ``circle`` / ``cvtColor`` call with its integer arguments, colours and order; ``draw_list`` below must
  reproduce those logs exactly.
  PARITY UNPINNED: the two OpenCV primitives themselves (``opencv-python`` is absent from ``/root/reference`` and from this
  image).  ``rasterize`` states them as remembered from OpenCV's ``drawing.cpp``: ``circle(r, filled)`` by the midpoint rule,
  which for r <= 3 is the set dx^2 + dy^2 <= r^2; ``line(thickness=3)`` = ``ThickLine``: a filled rectangle of half-width
  thickness / 2 + 1 / 2 = 2.0 px around the segment plus filled circles of radius 2 at both ends.  Integer arithmetic only.
"""
from __future__ import annotations

import numpy as np

LINE, CIRCLE, FLIP = 0, 1, 2
LINE_BGR, CIRCLE_BGR = (0, 0, 255), (0, 255, 0)


def scale_tracks(tracks, size, original_size, mode="inference"):
    """``tracks``: {id: [[x, y], ...]} (insertion order = draw order); ``size`` = [H, W]; ``original_size`` = (H0, W0[, 3])."""
    out = []
    for key in tracks:
        if mode == "inference":                                  # scripts/run_inference_vipseg_json_repro.py:431
            out.append([[int(p[0] * (size[1] / original_size[1])), int(p[1] * (size[0] / original_size[0]))] for p in tracks[key]])
        else:                                                    # utils/dataset.py:750
            out.append([[int(p[0] / original_size[1] * size[1]), int(p[1] / original_size[0] * size[0])] for p in tracks[key]])
    return out


def draw_list(scaled, start, end, mode="inference"):
    """Per map ``t`` in [start, end): the ordered calls ``(kind, x0, y0, x1, y1, c0, c1, c2, w)`` - LINE from point t to t+1
    (thickness w = 3), CIRCLE at point t+1 (x1 = y1 = 0, radius w = 3), FLIP (channel reversal) after every track ("dataset") or
    once after all tracks ("inference")."""
    maps = []
    for t in range(start, end):
        calls = []
        for tr in scaled:
            calls.append((LINE, tr[t][0], tr[t][1], tr[t + 1][0], tr[t + 1][1], *LINE_BGR, 3))
            calls.append((CIRCLE, tr[t + 1][0], tr[t + 1][1], 0, 0, *CIRCLE_BGR, 3))
            if mode == "dataset":
                calls.append((FLIP, 0, 0, 0, 0, 0, 0, 0, 0))
        if mode == "inference":
            calls.append((FLIP, 0, 0, 0, 0, 0, 0, 0, 0))
        maps.append(calls)
    return maps


def _disc(img, cx, cy, r, color):
    h, w = img.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    img[(xs - cx) ** 2 + (ys - cy) ** 2 <= r * r] = color


def _thick_line(img, x0, y0, x1, y1, color):
    h, w = img.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    dx, dy = x1 - x0, y1 - y0
    l2 = dx * dx + dy * dy
    if l2 > 0:
        vx, vy = xs - x0, ys - y0
        cross, dot = dx * vy - dy * vx, dx * vx + dy * vy
        img[(cross * cross <= 4 * l2) & (dot >= 0) & (dot <= l2)] = color
    _disc(img, x0, y0, 2, color)
    _disc(img, x1, y1, 2, color)


def rasterize(calls, size):
    """One map from its call list: uint8 ``[H, W, 3]`` in the array's final channel order (RGB after the flips)."""
    img = np.zeros((size[0], size[1], 3), dtype=np.uint8)
    for kind, x0, y0, x1, y1, c0, c1, c2, w in calls:
        if kind == LINE:
            _thick_line(img, x0, y0, x1, y1, (c0, c1, c2))
        elif kind == CIRCLE:
            _disc(img, x0, y0, w, (c0, c1, c2))
        else:
            img = img[..., ::-1].copy()
    return img


def trajectory_maps(tracks, size, original_size, num_frames=14, mode="inference", start=0):
    """The control maps as the pipeline wants them: float32 ``[num_frames, 3, H, W]`` in [-1, 1] - ``num_frames - 1`` drawn maps
    and a black last one (``scripts/...:446-447``) after ``image_processor.preprocess`` (x / 255 * 2 - 1)."""
    scaled = scale_tracks(tracks, size, original_size, mode)
    maps = [rasterize(c, size) for c in draw_list(scaled, start, start + num_frames - 1, mode)]
    maps.append(np.zeros_like(maps[0]))
    arr = np.stack(maps).astype(np.float32) / 255.0
    return np.ascontiguousarray(arr.transpose(0, 3, 1, 2)) * 2.0 - 1.0
