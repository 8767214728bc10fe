"""Storage-precision ladder for the oracle.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

The reference runs its hot path with ``.half()`` weights (``scripts/train_svd_traj_VIPSeg_14.py:1482-1492``,
``start_ft.sh``: ``--mixed_precision fp16``): every tensor an op hands to the next is an fp16 tensor, while each
op's arithmetic accumulates in fp32.  The oracle's arithmetic is fp32 throughout; ``q()`` marks every place where such
a hand-over happens, so that the same restatement can be run at three storage precisions:

``fp32``        ``q`` is the identity (the default; bit-identical to an oracle without the marks - the goldens under
                ``tests/golden/`` were generated in this mode and still pin it)
``fp16``        every marked tensor is rounded to fp16 (RNE) and widened again: "the reference in fp16" - one
                rounding per op output, fp32 math inside the op
``fp16-fused``  only the marks flagged ``store=True`` round: the subset of hand-overs that the MI355X path keeps as
                fp16 tensors in HBM (or feeds to an MFMA as fp16).  Everything the HIP kernels fuse into one epilogue
                (bias + time-embedding row + residual + blend + scale) is ONE rounding here too.

Three distances on the same inputs then separate the dtype from the implementation (``tools/parity_report.py``):
``HIP <-> fp16-fused`` (implementation: accumulation order, transcendental approximations, the attention kernel's
fp16 P operand), ``fp16-fused <-> fp32`` and ``fp16 <-> fp32`` (pure dtype), ``HIP <-> fp32`` (what the tests assert).
"""
from __future__ import annotations

import contextlib
import os

import torch

MODES = ("fp32", "fp16", "fp16-fused")
# mirror of posetraj_amd.ops.WIDE_STREAM for the "fp16-fused" storage model: which stream stores are fp16 pairs
# ("sc" shortcut conv, "xs" spatial resnet output, "rb" resblock output, "tr" transformer output, "ds" downsampler)
# Measured on the tiny nets (U-Net forward, fp16-fused vs fp32): none 1.15e-3; sc+xs+rb 7.8e-4; all five 7.7e-4.
# PT_WIDE_KINDS: the same A/B knob posetraj_amd.ops reads (profiles/r02/parity_wide_kinds.txt)
WIDE_STREAM = frozenset(k for k in os.environ.get("PT_WIDE_KINDS", "sc,xs,rb").split(",") if k)
_mode = "fp32"


def mode() -> str:
    return _mode


@contextlib.contextmanager
def storage(m: str):
    """``with storage("fp16"): y = net(x)``"""
    global _mode
    if m not in MODES:
        raise ValueError(f"unknown storage mode {m!r}; one of {MODES}")
    prev, _mode = _mode, m
    try:
        yield
    finally:
        _mode = prev


def q(x: torch.Tensor, store: bool = False, wide=None) -> torch.Tensor:
    """Hand-over point of a tensor between two ops (``store=True``: the MI355X path also materialises it in fp16;
    ``wide="<kind>"``: it materialises it as an fp16 PAIR ``hi = fp16(x)``, ``lo = fp16(x - hi)`` - the residual-stream tensors,
    ``posetraj_amd/ops.py: WIDE_STREAM`` - which ``fp16-fused`` models as that pair and ``fp16`` as plain fp16)."""
    if _mode == "fp32" or not torch.is_floating_point(x):
        return x
    if wide and _mode == "fp16-fused" and wide in WIDE_STREAM:
        hi = x.to(torch.float16).to(x.dtype)
        return hi + (x - hi).to(torch.float16).to(x.dtype)
    if _mode == "fp16" or store:
        return x.to(torch.float16).to(x.dtype)
    return x
