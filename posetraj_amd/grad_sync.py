"""Data-parallel gradient exchange for ``ControlNetTrainer``: the role ``accelerate``'s DDP wrapper plays in the reference
(``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1117-1119`` ``accelerator.prepare(..., controlnet)``; every rank trains on
its own clips, gradients are averaged before ``optimizer.step()``, and inside an accumulation cycle only the last micro-batch
synchronises).

The trainer's gradients already live in ONE flat fp32 buffer, so the exchange is a handful of large all-reduces over contiguous
slices of it - RCCL over xGMI (``torch.distributed`` backend "nccl" on ROCm), per-link bound, hence few and big - and each
slice is sent as soon as the reverse pass has produced every gradient inside it, while the rest of the backward still runs
(the collective is enqueued behind the current stream's work and ``wait()``-ed before the optimizer).  The reverse pass reaches
the parameters roughly in the reverse of their order in the buffer (zero-convs first, ``conv_in`` last), so buckets complete
from the tail.  Which parameters receive a gradient at all is learned from the first synchronised step (the single-key
cross-attentions leave ``to_q`` / ``to_k`` / ``norm2`` without one): that step sends everything at the end, later ones overlap.
With a world size of one every method is a no-op.

Two rules keep the ranks paired whatever their local state is.  (1) Buckets leave in ONE order on every rank - from
the last to the first, the order the reverse pass completes them in - and a complete bucket waits for the ones above it: WHEN a
bucket is sent is rank-local (learned set, signature, a held newcomer), WHICH collective is the n-th of a step is not, so two
equal-sized slices can never be summed crosswise.  (2) What a step learned is agreed on before it is trusted: ``finish()``
all-reduces a checksum of (signature, gradient-producing set) while the host waits anyway, and if any rank saw another set every
rank forgets its own and the next step sends at the end again.  ``signature`` must still be the same on all ranks of a step for
the overlap to happen; a rank that reaches a parameter AFTER its bucket has left raises (after which ``finish()`` only collects
what is in flight), which the agreement makes impossible as long as equal signatures mean equal graphs.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


_HOLD = object()                                            # in a bucket's pending set: "do not send before finish()"


class GradientBuckets:
    def __init__(self, flat_grad: torch.Tensor, spans: Dict[str, Tuple[int, int]], group=None, bucket_bytes: int = 256 << 20):
        """``spans``: parameter name -> (start, numel) inside ``flat_grad``."""
        self.flat, self.group = flat_grad, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        per = max(1, bucket_bytes // flat_grad.element_size())
        n = flat_grad.numel()
        self.bounds: List[Tuple[int, int]] = [(a, min(a + per, n)) for a in range(0, n, per)]
        self._buckets_of: Dict[str, List[int]] = {}
        for name, (start, numel) in spans.items():
            if numel > 0:
                self._buckets_of[name] = list(range(start // per, (start + numel - 1) // per + 1))
        self._expected: Optional[List[set]] = None          # per bucket: the parameters that produced a gradient last time
        self._signature = None                              # what the caller said about the step the set was learned from
        self._seen: List[set] = []
        self._pending: List[set] = []
        self._sent: List[bool] = []
        self._work = []
        self._active = False
        self._next = -1                                     # the highest bucket that has not left: the only one that may
        self._ready: List[bool] = []                        # complete and not yet sent (waiting for the buckets above it)
        self.launched_early = 0                             # buckets sent before finish() in the last cycle (overlap achieved)
        self.streams = []                                   # device streams that write gradients (the trainer's main and side streams)

    def begin(self, signature=None) -> None:
        """Start of the reverse pass whose gradients are final (the last micro-batch of an accumulation cycle).
        ``signature``: anything hashable that decides WHICH parameters this step's graph reaches (the trainer passes
        ``(use_spatial, camera_cond is not None)``).  The early sends trust the set learned from the last synchronised step; when
        the signature differs from that step's the set is dropped and this step sends everything at the end again."""
        if self.world == 1:
            return
        if signature != self._signature:
            self._expected, self._signature = None, signature
        self._collect()                                     # (in flight from a step that raised: never under the next step's writes)
        nb = len(self.bounds)
        self._seen = [set() for _ in range(nb)]
        self._pending = [set(s) for s in self._expected] if self._expected is not None else [set() for _ in range(nb)]
        self._sent = [False] * nb
        self._ready = [False] * nb
        self._next = nb - 1
        self._active = True
        self.launched_early = 0

    def _collect(self) -> None:
        for w in self._work:
            w.wait()
        self._work = []

    def _send(self, b: int) -> None:
        a, e = self.bounds[b]
        if self.flat.is_cuda and self.streams:              # the bucket's gradients come from kernels on several streams (weight
            cur = torch.cuda.current_stream()               # gradients run beside the data gradients): the collective is enqueued
            for st in self.streams:                         # behind all of them, whichever stream marked the last parameter
                if st is not None and st != cur:
                    cur.wait_stream(st)
        assert b == self._next, (b, self._next)             # rule (1): one order on every rank
        self._work.append(dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._sent[b] = True
        self._next = b - 1

    def mark_ready(self, name: str) -> None:
        """The gradient of ``name`` is complete for this step."""
        if self.world == 1 or not self._active:
            return
        for b in self._buckets_of.get(name, ()):
            if self._sent[b]:
                # The bucket has left (or is leaving): this gradient was written into a buffer that is being all-reduced in place,
                # or after it - this rank's contribution is lost or torn, the ranks would diverge silently.  Fail instead; the
                # cycle is over (nothing more is sent) and finish() / the next begin() only collect what is in flight.
                self._active = False
                raise RuntimeError(
                    f"GradientBuckets: the gradient of {name!r} was completed after its bucket had been all-reduced.  The early "
                    "sends trust the set of gradient-producing parameters learned from the previous synchronised step; this step "
                    "reached a parameter outside it.  Pass begin(signature=...) a value that changes whenever the step's graph "
                    "does (ControlNetTrainer does), or call reset() before such a step.")
            self._seen[b].add(name)
            if self._expected is not None:
                if name not in self._expected[b]:
                    self._pending[b].add(_HOLD)             # a newcomer: its bucket waits for finish(), which learns it (_seen)
                self._pending[b].discard(name)
                if not self._pending[b] and self._expected[b]:
                    self._ready[b] = True
                    while self._next >= 0 and self._ready[self._next]:       # this bucket and every complete one waiting below it
                        self._send(self._next)
                        self.launched_early += 1

    def reset(self) -> None:
        """Forget which parameters produce a gradient: the next synchronised step sends every bucket at the end and re-learns."""
        self._expected = None

    def finish(self) -> None:
        """Send what has not been sent, wait for everything.  ``flat`` then holds the SUM over ranks (the trainer folds the
        division by the world size into its un-scaling)."""
        if self.world == 1:
            return
        if not self._active:
            self._collect()
            return
        while self._next >= 0:
            self._send(self._next)
        # rule (2): the set this step produced is learned only if every rank produced the same one under the same signature
        text = repr((self._signature, [sorted(s) for s in self._seen])).encode()
        h = float(zlib.crc32(text))                         # < 2^32: exact in fp32's neighbour fp64, and in the flat buffer's device
        agree = torch.tensor([h, -h], dtype=torch.float64, device=self.flat.device)
        self._work.append(dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        self._collect()
        same = float(agree[0]) == -float(agree[1])          # max == min
        self._expected = self._seen if same else None
        self._active = False


def broadcast_parameters(flat: torch.Tensor, group=None, src: int = 0) -> None:
    """Every rank starts from rank ``src``'s parameters (what DDP does at wrap time)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
