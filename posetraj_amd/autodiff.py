"""Reverse-mode tape over the HIP kernels for the ControlNet training step (SURVEY 8f4;
``/root/reference/scripts/train_svd_traj_VIPSeg_14.py:1408-1425``: ``accelerator.backward(loss)``, ``optimizer.step()``).

The reference leans on ``torch.autograd`` over cuDNN / cuBLAS kernels; here every primitive of the training-mode forward
records a closure that runs its backward on this library's own kernels: data gradients of convolutions / linear layers are
``pt_igemm_f16`` over a transposed, tap-flipped pack; weight gradients and attention's backward are ``pt_gemm_f16``; norms,
activations and reductions are the kernels of ``csrc/backward.hip``.  PyTorch holds the buffers and does the plumbing (allocation,
zero fills, copies, dtype casts of the flat store and of a few row vectors): there is no autograd graph, every product, norm,
activation, reduction and the optimizer update is a kernel of this library, and - as everywhere in this package - there is no CPU path.

Precision follows ``--mixed_precision="fp16"`` of the reference's launch scripts (``start_ft.sh``): fp32 master parameters and
gradients, fp16 activations and activation gradients, the loss scaled before the reverse pass (``GradScaler`` semantics).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import hip, ops
from .ops import Packed, _ptr, _stream
from .packing import pack_conv2d, pack_conv_t3, pack_linear


# ------------------------------------------------------------------------------------------------- tape
class Var:
    """A tensor of the training-mode forward (fp16 ``[rows, C]``) and the gradient that flows back to it."""
    __slots__ = ("v", "g", "need")

    def __init__(self, v: torch.Tensor, need: bool = True):
        self.v, self.g, self.need = v, None, need


class Tape:
    def __init__(self):
        self._ops: List[Callable[[], None]] = []

    def record(self, fn: Callable[[], None]) -> None:
        self._ops.append(fn)

    def backward(self) -> None:
        for fn in reversed(self._ops):
            fn()
        self._ops.clear()


WGRAD_STREAM: Optional[torch.cuda.Stream] = None      # set by the trainer: weight gradients run beside the data gradients


def _beside(fn: Callable[[], None], *tensors: torch.Tensor) -> None:
    """Run ``fn`` (a layer's weight / bias gradient) on the side stream: it only needs tensors that exist by now and nobody
    waits for its result before the optimizer, while the data gradient of the same layer - on the main stream - is what the
    rest of the reverse pass waits for.  The low-resolution layers fill a fraction of the chip each (PMC: 13-33 % of the SIMD
    cycles have a resident wave), so the two overlap."""
    if WGRAD_STREAM is None:
        fn()
        return
    WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
    for t in tensors:
        t.record_stream(WGRAD_STREAM)
    with torch.cuda.stream(WGRAD_STREAM):
        fn()


def _acc(var: Optional[Var], g: torch.Tensor) -> None:
    """``var.g += g``; gradients are never modified in place (a tensor may be the gradient of two variables)."""
    if var is None or not var.need:
        return
    if var.g is None:
        var.g = g
    else:
        var.g = ops.axpy(var.g, g.view(var.g.shape), 1.0)


# ------------------------------------------------------------------------------------------------- kernel wrappers
def gemm(A, B, Cm, M, N, K, sa, sb, sc, *, nb=(1, 1, 1), ba=(0, 0, 0), bb=(0, 0, 0), bc=(0, 0, 0), alpha=1.0, out_mode=0, splits=1,
         gather=None):
    """``pt_gemm_f16``; ``A`` / ``B`` / ``Cm``: (tensor, element offset).  sa = (sa_m, sa_k), sb = (sb_k, sb_n), sc = (sc_m, sc_n)."""
    p = hip.GemmParams()
    (ta, oa), (tb, ob), (tc, oc) = A, B, Cm
    if not (ta.is_cuda and tb.is_cuda and tc.is_cuda) or ta.dtype != torch.float16 or tb.dtype != torch.float16 or \
            tc.dtype != (torch.float16 if out_mode == 0 else torch.float32):
        raise RuntimeError("posetraj_amd.gemm: fp16 operands and an fp16 (out_mode 0) / fp32 (out_mode 1-3) result on the GPU (no CPU path exists)")
    p.A, p.B = ta.data_ptr() + 2 * oa, tb.data_ptr() + 2 * ob
    p.C = tc.data_ptr() + tc.element_size() * oc
    p.M, p.N, p.K = M, N, K
    p.sa_m, p.sa_k = sa
    p.sb_k, p.sb_n = sb
    p.sc_m, p.sc_n = sc
    p.nb0, p.nb1, p.nb2 = nb
    p.ba0, p.ba1, p.ba2 = ba
    p.bb0, p.bb1, p.bb2 = bb
    p.bc0, p.bc1, p.bc2 = bc
    p.alpha, p.out_mode, p.splits = float(alpha), out_mode, splits
    if gather is not None:
        (p.g_H, p.g_W, p.g_OH, p.g_OW, p.g_KH, p.g_KW, p.g_stride, p.g_pad_h, p.g_pad_w, p.g_ld) = gather
    hip.check(hip.lib().pt_gemm_f16(C.byref(p), _stream()), "pt_gemm_f16")
    if GEMM_LOG is not None:
        GEMM_LOG.append((M, N, K, nb[0] * nb[1] * nb[2], "T" if sa[0] == 1 else "N", "T" if sb[1] == 1 else "N", out_mode, splits, gather is not None))


GEMM_LOG = None             # tools/train_step_bench.py --gemm-table: one tuple per pt_gemm_f16 launch
GEMV_ROWS = 16


def _few_rows(x: torch.Tensor, pw: Packed) -> bool:
    return x.dim() == 2 and x.shape[0] <= GEMV_ROWS and pw.K % 8 == 0 and x.stride(0) % 8 == 0 and not pw.geglu and not pw.silu


def gemv(x: torch.Tensor, pw: Packed, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``pt_gemv_f16``: a linear layer over at most 16 rows."""
    M = x.shape[0]
    out = torch.empty((M, pw.N), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().pt_gemv_f16(x.data_ptr(), x.stride(0), M, pw.w.data_ptr(), pw.Kpad, pw.K, pw.N, _ptr(pw.bias), _ptr(res),
                                    0 if res is None else res.stride(0), out.data_ptr(), pw.N, _stream()), "pt_gemv_f16")
    return out


def colsum(dy: torch.Tensor, rows_per_seg: int, nseg: int, out: torch.Tensor, ncols: Optional[int] = None) -> None:
    """``out[seg, c] += sum_rows dy`` (fp32 ``out``)."""
    hip.check(hip.lib().pt_colsum_f16(dy.data_ptr(), rows_per_seg, nseg, ncols or dy.shape[-1], dy.stride(-2), out.data_ptr(), _stream()),
              "pt_colsum_f16")


def _split_k(tiles: int, K: int) -> int:
    """Split-K factor of a weight gradient: enough workgroups for 256 CUs, at least 512 pixels per split.  A handful of tiles
    over millions of pixels (the condition encoder's 16 x 16 x 9 weights against 2.6 M pixels) is a latency chain per workgroup:
    four times as many, shorter, chains there."""
    return int(max(1, min((4096 if tiles <= 16 else 1024) // max(tiles, 1), K // 512)))


# ------------------------------------------------------------------------------------------------- parameters
class ParamStore:
    """The trainable network's parameters as ONE flat fp32 buffer (+ gradient and Adam moments): named views for the layers,
    one launch for the optimizer.  ``version`` counts optimizer steps: layers re-pack their fp16 operands when it moves.

    Convolution weights are kept TAP-MAJOR, ``[kh kw][Co][Ci]`` instead of torch's ``[Co][Ci][kh][kw]``: a weight gradient is
    then one ``[Co, Ci]`` matrix per tap, written with unit stride by ``pt_gemm_f16`` (in torch's layout neighbouring input
    channels lie 36 bytes apart and the low-resolution layers became bound by scattered fp32 atomics), and the pack kernel
    reads whole rows.  ``value()`` / ``gradient()`` return views in torch's shape (permuted strides), so callers never see it;
    AdamW is element-wise and does not care."""

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        self.names = list(sd)
        self.offsets, n = {}, 0
        for k in self.names:
            self.offsets[k] = n
            n += (sd[k].numel() + 7) // 8 * 8          # 16-byte aligned in the fp16 mirror too
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.shapes = {k: tuple(sd[k].shape) for k in self.names}
        for k in self.names:
            self.value(k).copy_(sd[k].to(device=device, dtype=torch.float32))
        self.version, self.trainable = 0, True
        self.on_grad_ready = None
        self.flat16 = torch.empty(n, dtype=torch.float16, device=device)      # fp16 mirror: norm weights / biases are read from it
        self._mirror_version = -1
        self._scalar_names = [k for k in self.names if sd[k].numel() == 1]
        self._scalar_index = torch.tensor([self.offsets[k] for k in self._scalar_names], dtype=torch.int64, device=device)
        self._scalars = None
        # the same scalars for launches that cannot take a host float (the step replayed as a hipGraph): sigmoid(value) of every
        # one-element parameter - they are all AlphaBlender mix factors - refreshed by ONE launch per step
        self.layers = []                              # the Dense layers built over this store (they register themselves)
        self.device_scalars = False
        self.alphas = torch.zeros(max(1, len(self._scalar_names)), dtype=torch.float32, device=device)
        self._alpha_slot = {k: i for i, k in enumerate(self._scalar_names)}

    def refresh_packs(self) -> int:
        """Re-write the fp16 packs of every layer whose packs exist, for the CURRENT version of the parameters, on the current stream:
        what ``Dense.packs()`` would do lazily, layer by layer, in the middle of the next forward (~400 launches of ~9 us in its
        serial chain: 3.5 ms of a 97 ms step).  The trainer calls this right behind AdamW on a stream of its own, so the launches run
        while the host stages the next step's inputs and beside the frozen encoder that opens it.  Returns the number of layers."""
        n = 0
        for L in self.layers:
            if L._packs is not None and L._packs[0] != self.version:
                L._refresh(L._packs[1], L._packs[2])
                L._packs = (self.version, L._packs[1], L._packs[2])
                n += 1
        return n

    def refresh_alphas(self) -> None:
        """``alphas[i] = sigmoid(scalar i)`` from the fp32 master buffer; enqueued at the start of every step in device-scalar mode."""
        if self._scalar_names:
            hip.check(hip.lib().pt_sigmoid_gather_f32(self.flat.data_ptr(), self._scalar_index.data_ptr(), len(self._scalar_names),
                                                      self.alphas.data_ptr(), _stream()), "pt_sigmoid_gather_f32")

    def alpha_ptr(self, k) -> int:
        return self.alphas.data_ptr() + 4 * self._alpha_slot[k]

    def half_view(self, k):
        """The parameter in the fp16 mirror (refreshed once per optimizer step: one cast over the flat buffer)."""
        if self._mirror_version != self.version:
            self.flat16.copy_(self.flat)
            self._mirror_version = self.version
        return self._view(self.flat16, k)

    def scalar(self, k) -> float:
        """A one-element parameter's value on the host; all of them travel in ONE copy per optimizer step."""
        if self._scalars is None or self._scalars[0] != self.version:
            vals = self.flat[self._scalar_index].cpu().tolist() if self._scalar_names else []
            self._scalars = (self.version, dict(zip(self._scalar_names, vals)))
        return self._scalars[1][k]

    def raw(self, buf, k):
        """The parameter's flat slice of ``buf`` (tap-major for convolution weights)."""
        o = self.offsets[k]
        n = 1
        for s in self.shapes[k]:
            n *= s
        return buf[o:o + n]

    def layout(self, k):
        """``(T, Co, Ci)`` of a weight (T = taps; 1 for a linear layer), None for vectors / scalars."""
        shape = self.shapes[k]
        if len(shape) < 2:
            return None
        T = 1
        for s in shape[2:]:
            T *= s
        return T, shape[0], shape[1]

    def _view(self, buf, k):
        shape = self.shapes[k]
        lay = self.layout(k)
        flat = self.raw(buf, k)
        if lay is None or lay[0] == 1:
            return flat.view(shape)
        T, Co, Ci = lay
        return flat.view(T, Co, Ci).permute(1, 2, 0).unflatten(2, shape[2:])

    def value(self, k):
        return self._view(self.flat, k)

    def gradient(self, k):
        return self._view(self.grad, k)

    def has(self, k):
        return k in self.offsets

    def stacked(self, names: Tuple[str, ...], buf=None):
        """Adjacent 2-D parameters of equal width as one ``[sum rows, K]`` view (to_q | to_k | to_v)."""
        buf = self.flat if buf is None else buf
        o0 = self.offsets[names[0]]
        rows, K = 0, self.shapes[names[0]][1]
        for k in names:
            if self.offsets[k] != o0 + rows * K or self.shapes[k][1] != K:
                raise RuntimeError(f"parameters {names} are not adjacent in the store")
            rows += self.shapes[k][0]
        return buf[o0:o0 + rows * K].view(rows, K)

    def export(self, buf):
        """``buf`` (the values, a moment buffer ...) as name -> tensor in torch's shapes and memory order."""
        return {k: self._view(buf, k).detach().contiguous().clone() for k in self.names}

    def load(self, buf, sd):
        """The inverse of ``export``: strict in names and shapes."""
        missing, extra = [k for k in self.names if k not in sd], [k for k in sd if k not in self.offsets]
        if missing or extra:
            raise KeyError(f"ParamStore.load: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
        for k in self.names:
            if tuple(sd[k].shape) != self.shapes[k]:
                raise ValueError(f"ParamStore.load: {k} has shape {tuple(sd[k].shape)}, expected {self.shapes[k]}")
            self._view(buf, k).copy_(sd[k].to(device=buf.device, dtype=buf.dtype))

    def state_dict(self):
        return self.export(self.flat)

    def zero_grad(self):
        self.grad.zero_()

    def spans(self):
        """name -> (start, numel) inside the flat buffers (grad_sync.GradientBuckets)."""
        out = {}
        for k in self.names:
            n = 1
            for s in self.shapes[k]:
                n *= s
            out[k] = (self.offsets[k], n)
        return out

    def grad_ready(self, *names):
        """The reverse pass has finished the gradients of ``names`` (every trainable layer is used once per step): the
        data-parallel exchange may send the buckets they complete."""
        if self.on_grad_ready is not None:
            for k in names:
                if k is not None:
                    self.on_grad_ready(k)


class FrozenParams:
    """A frozen network's tensors (any float dtype, moved to the device on first use); no gradients."""
    trainable, version = False, 0

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        self._sd, self._dev, self._cache = sd, device, {}

    def has(self, k):
        return k in self._sd

    def value(self, k):
        if k not in self._cache:
            self._cache[k] = self._sd[k].to(device=self._dev)
        return self._cache[k]

    def gradient(self, k):
        return None

    def grad_ready(self, *names):
        pass

    def half_view(self, k):
        return self.value(k).to(torch.float16).contiguous()

    def scalar(self, k) -> float:
        return float(self.value(k).detach().float().cpu().reshape(-1)[0])

    def stacked(self, names, buf=None):
        return torch.cat([self.value(k) for k in names], 0)

    def release(self):
        self._cache.clear()


class Dense:
    """A linear / convolution layer in training mode: fp16 packs for the forward (``[N, K]``) and the data gradient (the
    transposed, tap-flipped weight), rebuilt from the fp32 master when the optimizer moved it; weight / bias gradients."""

    def __init__(self, P, wname, bname=None, kind: str = "linear", stride: int = 1, padding: int = 0, stack: Optional[Tuple[str, ...]] = None,
                 kpad: Optional[int] = None, dgrad_cols: Optional[int] = None):
        """``kpad`` (linear layers): the input arrives zero-padded to ``kpad`` columns (the camera twin's ``cc_projection`` reads
        ``C + 12`` channels padded to a multiple of 8); ``dgrad_cols``: only the first that many input columns need a gradient."""
        self.P, self.wname, self.kind, self.stride, self.padding, self.stack = P, wname, kind, stride, padding, stack
        self.kpad, self.dgrad_cols = kpad, dgrad_cols
        self.bname = bname if (bname is not None and P.has(bname)) else None
        self._packs = None
        if isinstance(P, ParamStore):
            P.layers.append(self)                     # (ParamStore.refresh_packs re-writes every trainable layer's packs in one go)

    def weight(self):
        return self.P.stacked(self.stack) if self.stack else self.P.value(self.wname)

    def _refresh(self, f, t):
        """Rewrite both packs in place from the fp32 master (pt_pack_weight_f32): same buffers, same addresses."""
        src, (T, Co, Ci) = self._raw(self.P.flat)
        if self.kind == "linear":
            Ci, T = Ci * T, 1                         # a 1 x 1 convolution's weight used as a matrix
            cpf, cpt = Ci, Co
        elif self.kind == "conv":
            cpf, cpt = f.cin, t.cin
        else:
            cpf, cpt = Ci, Co
        b = None if self.bname is None else self.P.value(self.bname)
        L = hip.lib()
        hip.check(L.pt_pack_weight_f32(src.data_ptr(), Co, Ci, T, 0, _ptr(b), f.w.data_ptr(), f.Kpad, cpf, _ptr(f.bias), _stream()), "pt_pack_weight_f32")
        hip.check(L.pt_pack_weight_f32(src.data_ptr(), Co, Ci, T, 1, None, t.w.data_ptr(), t.Kpad, cpt, None, _stream()), "pt_pack_weight_f32")

    def _raw(self, buf):
        """The weight's flat (tap-major) slice of ``buf`` and its (T, Co, Ci); stacked layers: the adjacent matrices as one."""
        P = self.P
        if self.stack:
            m = P.stacked(self.stack, buf)
            return m, (1, m.shape[0], m.shape[1])
        return P.raw(buf, self.wname), P.layout(self.wname)

    def packs(self):
        if self._packs is not None and self._packs[0] != self.P.version and isinstance(self.P, ParamStore):
            self._refresh(self._packs[1], self._packs[2])
            self._packs = (self.P.version, self._packs[1], self._packs[2])
        if self._packs is None or self._packs[0] != self.P.version:
            w = self.weight().detach()
            b = None if self.bname is None else self.P.value(self.bname)
            dev = w.device
            if self.kind == "linear":
                w2 = w.reshape(w.shape[0], -1)
                wf = w2
                if self.kpad is not None and self.kpad > w2.shape[1]:
                    wf = torch.zeros((w2.shape[0], self.kpad), dtype=w2.dtype, device=dev)
                    wf[:, :w2.shape[1]] = w2
                f, t = pack_linear(wf, b, dev), pack_linear(w2.t(), None, dev)
                if self.dgrad_cols is not None:
                    t.N = self.dgrad_cols                       # rows beyond stay in the pack, unread
            elif self.kind == "conv":
                kh = w.shape[2]
                f = pack_conv2d(w, b, dev, stride=self.stride, padding=self.padding)
                t = pack_conv2d(w.flip(2, 3).transpose(0, 1), None, dev, stride=1, padding=kh - 1 - self.padding)
            else:
                f = pack_conv_t3(w, b, dev)
                t = pack_conv_t3(w.flip(2).transpose(0, 1), None, dev)
            self._packs = (self.P.version, f, t)
        return self._packs[1], self._packs[2]

    def split_dgrad_packs(self, C0: int):
        """Data-gradient packs of a layer fed by two concatenated sources: one per source (output rows [0, C0) and [C0, Ci))."""
        key = ("split", C0)
        if getattr(self, "_split", None) is None or self._split[0] != (self.P.version, key):
            w = self.weight().detach()
            dev = w.device
            if self.kind == "conv":
                kh = w.shape[2]
                wt = w.flip(2, 3).transpose(0, 1)
                mk = lambda ww: pack_conv2d(ww.contiguous(), None, dev, stride=1, padding=kh - 1 - self.padding)
            else:
                wt = w.reshape(w.shape[0], -1).t()
                mk = lambda ww: pack_linear(ww.contiguous(), None, dev)
            self._split = ((self.P.version, key), mk(wt[:C0]), mk(wt[C0:]))
        return self._split[1], self._split[2]

    def accumulate(self, x: torch.Tensor, dy: torch.Tensor, geom) -> None:
        """dW += dY^T X (gathered per tap; one ``[Co, Ci]`` matrix per tap in the store's tap-major layout), db += column sums
        of dY.  Split-K over the pixels with fp32 atomics when the tile count alone would not fill the chip, else plain +=."""
        if not self.P.trainable:
            return
        gw, (T, Co, Ci) = self._raw(self.P.grad)
        ldy, ldx = dy.stride(-2), x.stride(-2)
        if self.kind == "linear":
            Ci, K = Ci * T, dy.shape[0]
            tiles = ((Co + 127) // 128) * ((Ci + 127) // 128)
            sp = _split_k(tiles, K)
            gemm((dy, 0), (x, 0), (gw, 0), Co, Ci, K, (1, ldy), (ldx, 1), (Ci, 1), out_mode=2 if sp > 1 else 3, splits=sp)
        else:
            Nimg, H, W = geom
            if self.kind == "conv":
                KH, KW = self.P.shapes[self.wname][2], self.P.shapes[self.wname][3]
                OH = (H + 2 * self.padding - KH) // self.stride + 1
                OW = (W + 2 * self.padding - KW) // self.stride + 1
                gat = (H, W, OH, OW, KH, KW, self.stride, self.padding, self.padding, ldx)
            else:
                KH, KW, OH, OW = 3, 1, H, W
                gat = (H, W, OH, OW, 3, 1, 1, 1, 0, ldx)
            K = Nimg * OH * OW
            tiles = ((Co + 127) // 128) * ((Ci + 127) // 128) * T
            sp = _split_k(tiles, K)
            gemm((dy, 0), (x, 0), (gw, 0), Co, Ci, K, (1, ldy), (ldx, 1), (Ci, 1), nb=(1, 1, T), bc=(0, 0, Co * Ci), out_mode=2 if sp > 1 else 3,
                 splits=sp, gather=gat)
        if self.bname is not None:
            colsum(dy, dy.shape[0], 1, self.P.gradient(self.bname), ncols=Co)
        self.P.grad_ready(*(self.stack or (self.wname,)), self.bname)


class Affine:
    """Norm weight / bias: fp16 copies for the kernels, fp32 gradients."""

    def __init__(self, P, prefix):
        self.P, self.w, self.b = P, prefix + ".weight", prefix + ".bias"
        self._c = None

    def halves(self):
        if self._c is None or self._c[0] != self.P.version:
            self._c = (self.P.version, self.P.half_view(self.w), self.P.half_view(self.b))
        return self._c[1], self._c[2]

    def grads(self):
        if not self.P.trainable:
            return None, None
        return self.P.gradient(self.w), self.P.gradient(self.b)


class Mix:
    """AlphaBlender's ``mix_factor`` (one scalar): alpha = sigmoid(mix_factor) is the weight of the SPATIAL branch."""

    def __init__(self, P, name):
        self.P, self.name, self._c = P, name, None

    def alpha(self) -> float:
        if self._c is None or self._c[0] != self.P.version:
            self._c = (self.P.version, 1.0 / (1.0 + math.exp(-self.P.scalar(self.name))))
        return self._c[1]


# ------------------------------------------------------------------------------------------------- primitives
def dense(tape: Tape, x: Var, L: Dense, *, geom=None, res: Optional[Var] = None, x1: Optional[Var] = None, upsample2x: bool = False) -> Var:
    """``y = L(x [| x1]) (+ res)``.  ``geom = (Nimg, H, W)`` of the input for convolutions (``(B, F, S)`` for the temporal ones)."""
    if x1 is not None and L.P.trainable:
        raise RuntimeError("weight gradients of a two-source layer are not implemented (only the frozen U-Net's up blocks have them)")
    fwd, _ = L.packs()
    if geom is None and x1 is None and _few_rows(x.v, fwd):
        y = gemv(x.v, fwd, None if res is None else res.v)
    else:
        y = ops.igemm(x.v, fwd, geom=geom, x1=None if x1 is None else x1.v, upsample2x=upsample2x, res=None if res is None else res.v)
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        _acc(res, dy)
        if x1 is None and L.P.trainable:
            _beside(lambda: L.accumulate(x.v, dy, geom), x.v, dy)
        if not (x.need or (x1 is not None and x1.need)):
            return
        _, tp = L.packs()
        if geom is None:
            lin = lambda p: gemv(dy, p) if _few_rows(dy, p) else ops.igemm(dy, p)
            gx = [lin(tp)] if x1 is None else [lin(p) for p in L.split_dgrad_packs(x.v.shape[-1])]
        else:
            Nimg, H, W = geom
            if upsample2x:
                du = ops.igemm(dy, tp, geom=(Nimg, 2 * H, 2 * W))
                d = torch.empty((Nimg * H * W, du.shape[-1]), dtype=torch.float16, device=du.device)
                hip.check(hip.lib().pt_sumpool2x_f16(du.data_ptr(), Nimg, H, W, du.shape[-1], d.data_ptr(), _stream()), "pt_sumpool2x_f16")
                gx = [d]
            elif L.stride == 2:
                OH, OW = (H + 2 * L.padding - 3) // 2 + 1, (W + 2 * L.padding - 3) // 2 + 1
                Co = dy.shape[-1]
                z = torch.empty((Nimg * H * W, Co), dtype=torch.float16, device=dy.device)
                hip.check(hip.lib().pt_zero_insert2x_f16(dy.data_ptr(), Nimg, OH, OW, H, W, Co, z.data_ptr(), _stream()), "pt_zero_insert2x_f16")
                gx = [ops.igemm(z, tp, geom=(Nimg, H, W))]
            elif x1 is None:
                gx = [ops.igemm(dy, tp, geom=geom)]
            else:
                gx = [ops.igemm(dy, p, geom=geom) for p in L.split_dgrad_packs(x.v.shape[-1])]
        _acc(x, gx[0])
        if x1 is not None:
            _acc(x1, gx[1])

    tape.record(bwd)
    return out


def groupnorm(tape: Tape, x: Var, A: Affine, *, rows_per_sample: int, n_samples: int, eps: float, silu: bool, x1: Optional[Var] = None,
              groups: int = 32) -> Var:
    gm, bt = A.halves()
    y = ops.groupnorm(x.v, gm, bt, rows_per_sample=rows_per_sample, n_samples=n_samples, eps=eps, silu=silu,
                      x1=None if x1 is None else x1.v, groups=groups)
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        C0 = x.v.shape[-1]
        C1 = 0 if x1 is None else x1.v.shape[-1]
        rows = rows_per_sample * n_samples
        dx0 = torch.empty((rows, C0), dtype=torch.float16, device=dy.device)
        dx1 = None if x1 is None else torch.empty((rows, C1), dtype=torch.float16, device=dy.device)
        stat = torch.empty(4 * n_samples * groups, dtype=torch.float32, device=dy.device)
        dg, db = A.grads()
        hip.check(hip.lib().pt_groupnorm_bwd(x.v.data_ptr(), _ptr(None if x1 is None else x1.v), C0, C1, groups, rows_per_sample, n_samples, float(eps),
                                             gm.data_ptr(), bt.data_ptr(), 1 if silu else 0, dy.data_ptr(), dx0.data_ptr(), _ptr(dx1), _ptr(dg), _ptr(db),
                                             stat.data_ptr(), _stream()), "pt_groupnorm_bwd")
        A.P.grad_ready(A.w, A.b)
        _acc(x, dx0)
        if x1 is not None:
            _acc(x1, dx1)

    tape.record(bwd)
    return out


def layernorm(tape: Tape, x: Var, A: Affine, eps: float = 1e-5) -> Var:
    gm, bt = A.halves()
    out = Var(ops.layernorm(x.v, gm, bt, eps))

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        M, Cc = x.v.shape
        dx = torch.empty_like(x.v)
        dg, db = A.grads()
        rowstat = None if dg is None else torch.empty(2 * M, dtype=torch.float32, device=dy.device)
        hip.check(hip.lib().pt_layernorm_bwd(x.v.data_ptr(), M, Cc, gm.data_ptr(), float(eps), dy.data_ptr(), dx.data_ptr(), _ptr(dg), _ptr(db),
                                             _ptr(rowstat), _stream()), "pt_layernorm_bwd")
        A.P.grad_ready(A.w, A.b)
        _acc(x, dx)

    tape.record(bwd)
    return out


def silu(tape: Tape, x: Var) -> Var:
    out = Var(ops.silu(x.v))

    def bwd():
        dy, out.g = out.g, None
        if dy is None or not x.need:
            return
        dx = torch.empty_like(x.v)
        hip.check(hip.lib().pt_silu_bwd(x.v.data_ptr(), dy.data_ptr(), x.v.numel(), dx.data_ptr(), _stream()), "pt_silu_bwd")
        _acc(x, dx)

    tape.record(bwd)
    return out


def geglu(tape: Tape, h: Var) -> Var:
    """diffusers ``GEGLU``: ``hidden, gate = proj.chunk(2, dim=-1); hidden * gelu(gate)``."""
    M, two_i = h.v.shape
    I = two_i // 2
    y = torch.empty((M, I), dtype=torch.float16, device=h.v.device)
    hip.check(hip.lib().pt_geglu_f16(h.v.data_ptr(), M, I, y.data_ptr(), _stream()), "pt_geglu_f16")
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        dh = torch.empty_like(h.v)
        hip.check(hip.lib().pt_geglu_bwd(h.v.data_ptr(), dy.data_ptr(), M, I, dh.data_ptr(), _stream()), "pt_geglu_bwd")
        _acc(h, dh)

    tape.record(bwd)
    return out


def add(tape: Tape, a: Var, b: Var) -> Var:
    out = Var(ops.axpy(a.v, b.v, 1.0).view(a.v.shape))

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        _acc(a, dy)
        _acc(b, dy)

    tape.record(bwd)
    return out


def add_scaled_const(tape: Tape, const: torch.Tensor, r: Var, m: float) -> Var:
    """``const + m * r`` where ``const`` carries no gradient (a U-Net skip receiving its ControlNet residual m times)."""
    out = Var(ops.axpy(const.reshape(r.v.shape), r.v, float(m)))

    def bwd():
        dy, out.g = out.g, None
        if dy is not None:
            _acc(r, dy if m == 1 else ops.scale(dy, float(m)))

    tape.record(bwd)
    return out


def add_rowvec(tape: Tape, x: Var, vec: Var, rows_per_vec: int) -> Var:
    """``y[r] = x[r] + vec[r // rows_per_vec]`` (time-embedding rows, collapsed cross-attention, frame position embedding)."""
    rows, Cc = x.v.shape
    y = torch.empty_like(x.v)
    hip.check(hip.lib().pt_add_rowvec_f16(x.v.data_ptr(), vec.v.data_ptr(), rows, Cc, rows_per_vec, y.data_ptr(), _stream()), "pt_add_rowvec_f16")
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        _acc(x, dy)
        if vec.need:
            nseg = rows // rows_per_vec
            s = torch.zeros((nseg, Cc), dtype=torch.float32, device=dy.device)
            colsum(dy, rows_per_vec, nseg, s)
            _acc(vec, s.to(torch.float16))

    tape.record(bwd)
    return out


def blend(tape: Tape, a: Var, b: Var, M: Mix) -> Var:
    """AlphaBlender (``merge_strategy="learned_with_images"`` with an all-zero indicator): ``alpha a + (1 - alpha) b``."""
    y = torch.empty_like(a.v)
    if M.P.trainable and getattr(M.P, "device_scalars", False):
        # the weight changes with every optimizer step: read from device memory (ParamStore.alphas), so that the launch can be replayed
        # inside a captured hipGraph (ControlNetTrainer(use_graph=True)); the frozen U-Net's weights below are constants of the capture
        ap = M.P.alpha_ptr(M.name)
        L = hip.lib()
        hip.check(L.pt_lerp_f16_dev(a.v.data_ptr(), b.v.data_ptr(), ap, a.v.numel(), y.data_ptr(), _stream()), "pt_lerp_f16_dev")
        out = Var(y)

        def scaled(dy, one_minus):
            d = dy.contiguous()
            z = torch.empty_like(d)
            hip.check(L.pt_scale_f16_dev(d.data_ptr(), ap, one_minus, d.numel(), z.data_ptr(), _stream()), "pt_scale_f16_dev")
            return z

        def bwd_dev():
            dy, out.g = out.g, None
            if dy is None:
                return
            hip.check(L.pt_dot_diff_dev(dy.data_ptr(), a.v.data_ptr(), b.v.data_ptr(), dy.numel(), ap, M.P.gradient(M.name).data_ptr(), _stream()),
                      "pt_dot_diff_dev")
            M.P.grad_ready(M.name)
            _acc(a, scaled(dy, 0))
            _acc(b, scaled(dy, 1))

        tape.record(bwd_dev)
        return out
    al = M.alpha()
    hip.check(hip.lib().pt_lerp_f16(a.v.data_ptr(), b.v.data_ptr(), al, a.v.numel(), y.data_ptr(), _stream()), "pt_lerp_f16")
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        if M.P.trainable:                                       # d alpha / d mix = alpha (1 - alpha)
            hip.check(hip.lib().pt_dot_diff(dy.data_ptr(), a.v.data_ptr(), b.v.data_ptr(), dy.numel(), al * (1.0 - al),
                                            M.P.gradient(M.name).data_ptr(), _stream()), "pt_dot_diff")
            M.P.grad_ready(M.name)
        _acc(a, ops.scale(dy, al))
        _acc(b, ops.scale(dy, 1.0 - al))

    tape.record(bwd)
    return out


def concat_camera(tape: Tape, x: Var, geom, cam: torch.Tensor, cpad: int) -> Var:
    """``cat([x, camera R|T tiled over the positions])`` zero-padded to ``cpad`` channels (``controlnet_sdv_cam_infer.py:109-116``);
    the camera values are inputs: the gradient that comes back covers the first C channels and goes to ``x``."""
    N, H, W = geom
    out = Var(ops.concat_camera(x.v.view(N, H, W, -1), cam, cpad).view(N * H * W, cpad))

    def bwd():
        dy, out.g = out.g, None
        if dy is not None:
            _acc(x, dy)

    tape.record(bwd)
    return out


def add_rows(x: Var, r0: int, r1: int, dy: torch.Tensor) -> None:
    """``x.g[r0:r1] += dy`` on the current stream (the join of a deferred ``rows`` gradient)."""
    if not x.need:
        return
    x.g = torch.zeros_like(x.v) if x.g is None else x.g.clone()
    sl = x.g[r0:r1]
    hip.check(hip.lib().pt_axpy_f16(sl.data_ptr(), dy.data_ptr(), 1.0, sl.data_ptr(), sl.numel(), _stream()), "pt_axpy_f16")


def rows(tape: Tape, x: Var, r0: int, r1: int, defer: Optional[list] = None) -> Var:
    """``x[r0:r1]`` (whole rows: a frame of a ``[frames * S, C]`` tensor).  ``defer``: the backward only appends
    ``(x, r0, r1, gradient)`` to this list - for a branch that runs on another stream than ``x``'s other consumers; the caller
    joins with ``add_rows`` once both streams are in."""
    out = Var(x.v[r0:r1])

    def bwd():
        dy, out.g = out.g, None
        if dy is None or not x.need:
            return
        if defer is not None:
            defer.append((x, r0, r1, dy))
            return
        if x.g is None:
            x.g = torch.zeros_like(x.v)
        else:
            x.g = x.g.clone()
        sl = x.g[r0:r1]
        hip.check(hip.lib().pt_axpy_f16(sl.data_ptr(), dy.data_ptr(), 1.0, sl.data_ptr(), sl.numel(), _stream()), "pt_axpy_f16")

    tape.record(bwd)
    return out


def _attention_backward(qkv: torch.Tensor, dout: torch.Tensor, Cc: int, heads: int, hd: int, Sx: int, tok: int, batches, chunk_outer: int):
    """Backward of softmax(Q K^T / sqrt(d)) V for (outer, inner, head) batches addressed inside the fused projection ``qkv``
    (columns [Q | K | V], row pitch ld): token t of batch (o, i) is row ``o * so + i * si + t * tok``.  Scores are recomputed
    (``pt_gemm_f16`` + ``pt_softmax_rows``) for ``chunk_outer`` outer entries at a time."""
    n_outer, so, n_inner, si = batches
    ld, ldo = qkv.stride(0), dout.stride(0)
    dqkv = torch.empty_like(qkv)
    scale = hd ** -0.5
    dev = qkv.device
    for o0 in range(0, n_outer, chunk_outer):
        no = min(chunk_outer, n_outer - o0)
        nb = (no, n_inner, heads)
        nbt = no * n_inner * heads
        sc = torch.empty((nbt, Sx, Sx), dtype=torch.float32, device=dev)
        P = torch.empty((nbt, Sx, Sx), dtype=torch.float16, device=dev)
        row0 = o0 * so
        bq = (so * ld, si * ld, hd)                  # batch offsets inside qkv (elements)
        bo = (so * ldo, si * ldo, hd)
        bs = (n_inner * heads * Sx * Sx, heads * Sx * Sx, Sx * Sx)
        q0, k0, v0 = row0 * ld, row0 * ld + Cc, row0 * ld + 2 * Cc
        # S = scale Q K^T
        gemm((qkv, q0), (qkv, k0), (sc, 0), Sx, Sx, hd, (tok * ld, 1), (1, tok * ld), (Sx, 1), nb=nb, ba=bq, bb=bq, bc=bs, alpha=scale, out_mode=1)
        hip.check(hip.lib().pt_softmax_rows(sc.data_ptr(), nbt * Sx, Sx, Sx, P.data_ptr(), Sx, _stream()), "pt_softmax_rows")
        # dV = P^T dO
        gemm((P, 0), (dout, row0 * ldo), (dqkv, v0), Sx, hd, Sx, (1, Sx), (tok * ldo, 1), (tok * ld, 1), nb=nb, ba=bs, bb=bo, bc=bq)
        # dP = dO V^T  (into the score buffer)
        gemm((dout, row0 * ldo), (qkv, v0), (sc, 0), Sx, Sx, hd, (tok * ldo, 1), (1, tok * ld), (Sx, 1), nb=nb, ba=bo, bb=bq, bc=bs, out_mode=1)
        dS = torch.empty_like(P)
        hip.check(hip.lib().pt_softmax_bwd_rows(P.data_ptr(), Sx, sc.data_ptr(), Sx, nbt * Sx, Sx, dS.data_ptr(), Sx, _stream()), "pt_softmax_bwd_rows")
        # dQ = scale dS K ; dK = scale dS^T Q
        gemm((dS, 0), (qkv, k0), (dqkv, q0), Sx, hd, Sx, (Sx, 1), (tok * ld, 1), (tok * ld, 1), nb=nb, ba=bs, bb=bq, bc=bq, alpha=scale)
        gemm((dS, 0), (qkv, q0), (dqkv, k0), Sx, hd, Sx, (1, Sx), (tok * ld, 1), (tok * ld, 1), nb=nb, ba=bs, bb=bq, bc=bq, alpha=scale)
    return dqkv


ATTN_SCORE_BYTES = 2 << 30          # score-matrix memory (fp32) the recomputing attention backward holds at a time
FLASH_BACKWARD = True               # spatial attention: pt_attn_fwd_lse_f16 / pt_attn_bwd_f16 (False: recompute through pt_gemm_f16)


def attn_spatial(tape: Tape, qkv: Var, N: int, S: int, heads: int, hd: int) -> Var:
    """Self-attention over the S positions of each of N frames; ``qkv`` = fused projection ``[N * S, 3 C]``."""
    Cc = heads * hd
    ld = qkv.v.stride(0)
    flash = FLASH_BACKWARD and hd in (64, 128)
    lse = None
    if flash:
        ops.ensure_ready(qkv.v.device)
        y = torch.empty((N * S, Cc), dtype=torch.float16, device=qkv.v.device)
        lse = torch.empty((N * S, heads), dtype=torch.float32, device=qkv.v.device)
        p0 = qkv.v.data_ptr()
        hip.check(hip.lib().pt_attn_fwd_lse_f16(p0, ld, p0 + 2 * Cc, ld, p0 + 4 * Cc, ld, y.data_ptr(), Cc, N, S, S, heads, hd, hd ** -0.5,
                                                lse.data_ptr(), _stream()), "pt_attn_fwd_lse_f16")
    elif hd == 64:
        y = ops.attn_spatial(qkv.v, N, S, heads, hd)
    else:
        y = ops.attention(qkv.v[:, :Cc], qkv.v[:, Cc:2 * Cc], qkv.v[:, 2 * Cc:], N, S, S, heads, hd)
    out = Var(y)

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        if not flash:
            chunk = max(1, ATTN_SCORE_BYTES // (heads * S * S * 4))
            _acc(qkv, _attention_backward(qkv.v, dy, Cc, heads, hd, S, 1, (N, S, 1, 0), chunk))
            return
        dqkv = torch.empty_like(qkv.v)
        dot = torch.empty((N * S, heads), dtype=torch.float32, device=dy.device)
        p0, d0 = qkv.v.data_ptr(), dqkv.data_ptr()
        hip.check(hip.lib().pt_attn_bwd_f16(p0, ld, p0 + 2 * Cc, ld, p0 + 4 * Cc, ld, out.v.data_ptr(), Cc, dy.data_ptr(), dy.stride(0), lse.data_ptr(),
                                            dot.data_ptr(), d0, d0 + 2 * Cc, d0 + 4 * Cc, dqkv.stride(0), N, S, heads, hd, hd ** -0.5, _stream()),
                  "pt_attn_bwd_f16")
        _acc(qkv, dqkv)

    tape.record(bwd)
    return out


def attn_temporal(tape: Tape, qkv: Var, B: int, F: int, S: int, heads: int, hd: int) -> Var:
    """Self-attention over the F frames of each spatial position; token f of (clip b, position s) is row (b F + f) S + s."""
    Cc = heads * hd
    out = Var(ops.attn_temporal(qkv.v, B, F, S, heads, hd))

    def bwd():
        dy, out.g = out.g, None
        if dy is None:
            return
        if F == 1:                                   # one frame (the spatial-loss pass): softmax over one key is 1, out = V -> dV = dO, dQ = dK = 0
            dqkv = torch.zeros_like(qkv.v)
            dqkv[:, 2 * Cc:].copy_(dy)
            _acc(qkv, dqkv)
            return
        if FLASH_BACKWARD and F <= 16 and hd in (64, 128):
            dqkv = torch.empty_like(qkv.v)
            hip.check(hip.lib().pt_attn_temporal_bwd_f16(qkv.v.data_ptr(), qkv.v.stride(0), Cc, 2 * Cc, dy.data_ptr(), dy.stride(0), dqkv.data_ptr(),
                                                         dqkv.stride(0), B, F, S, heads, hd, hd ** -0.5, _stream()), "pt_attn_temporal_bwd_f16")
            _acc(qkv, dqkv)
            return
        _acc(qkv, _attention_backward(qkv.v, dy, Cc, heads, hd, F, S, (B, F * S, S, 1), 1))

    tape.record(bwd)
    return out
