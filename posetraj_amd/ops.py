"""Thin tensor-level wrappers over the C ABI.  PyTorch is used for device memory and the current stream only;
all arithmetic happens in ``libposetraj_hip.so``.  Every function requires CUDA(ROCm) fp16 tensors and raises
otherwise - there is no CPU or eager-PyTorch fallback."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import hip

# The residual stream as fp16 pairs (hi, lo): resblock / transformer outputs and the shortcut are stored as
# fp16(v) + fp16(v - fp16(v)); GEMM operands and norms read the high half (the plain fp16 tensor), residual adds read
# both.  Removes the one-rounding-per-block random walk that carries 0.98e-3 of the U-Net's 1.08e-3 rel-L2.
WIDE_KINDS = frozenset(("sc", "xs", "rb"))                       # shortcut conv, spatial resnet output, resblock output
WIDE_STREAM = os.environ.get("PT_WIDE_STREAM", "1") != "0"      # (0: A/B of its cost, tools/ab_bench.py)


# Explicit-destination writes (``igemm(out=...)``) per buffer address.  A wide-stream tensor remembers the count its buffer had
# when its low half was produced (``lo_gen``); a later in-place write into the high half makes the pair stale, and the next use
# of it as a residual raises instead of silently adding a low half that no longer belongs to the high one.
_inplace_writes = {}


def _attach_lo(hi: torch.Tensor, lo: torch.Tensor) -> None:
    hi.lo = lo
    hi.lo_gen = _inplace_writes.get(hi.data_ptr(), 0)


def wview(t: torch.Tensor, *shape) -> torch.Tensor:
    """``t.view(*shape)`` that keeps the low half of a wide-stream tensor attached."""
    v = t.view(*shape)
    lo = getattr(t, "lo", None)
    if lo is not None:
        v.lo = lo.view(*shape)
        v.lo_gen = getattr(t, "lo_gen", 0)
    return v


def drop_lo(t: torch.Tensor) -> None:
    """Forget the low half of a wide-stream tensor whose high half was just overwritten in place."""
    if hasattr(t, "lo"):
        del t.lo
    if hasattr(t, "lo_gen"):
        del t.lo_gen


_zero_pages = {}          # device index -> the 256-byte zero page registered with the library for that device


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device as an integer handle.  Through torch's C entry points when they exist:
    ``torch.cuda.current_stream()`` builds a Stream object and re-checks ``is_available()`` (an ``os.environ`` lookup) on every
    call - 12 us, 4 500 times per training step (tools/micro/train_host_profile.py)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, name: str, dtype=torch.float16):
    if not t.is_cuda:
        raise RuntimeError(f"posetraj_amd: `{name}` must live on the GPU (no CPU path exists); got {t.device}")
    if t.dtype != dtype:
        raise RuntimeError(f"posetraj_amd: `{name}` must be {dtype}; got {t.dtype}")


def ensure_ready(device) -> None:
    """Allocates the zero page padded convolution taps read from and registers it with the library."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _zero_pages:
        with torch.cuda.device(idx):
            z = torch.zeros(256, dtype=torch.uint8, device=device)
            hip.check(hip.lib().pt_set_zero_page(z.data_ptr()), "pt_set_zero_page")
        _zero_pages[idx] = z


@dataclass
class Packed:
    """A pre-packed weight: fp16 ``w [Npad, Kpad]`` (K ordered ky, kx, ci), optional fp16 ``bias [Npad]``."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    K: int
    KH: int = 1
    KW: int = 1
    stride: int = 1
    pad_h: int = 0
    pad_w: int = 0
    cin: int = 0          # channels per tap (after padding to a multiple of 8)
    geglu: bool = False
    silu: bool = False

    @property
    def Kpad(self):
        return self.w.shape[1]

    @property
    def n_out(self):
        return self.N // 2 if self.geglu else self.N


def igemm(x0: torch.Tensor, pw: Packed, *, x1: Optional[torch.Tensor] = None, geom=None, upsample2x: bool = False,
          res: Optional[torch.Tensor] = None, vec: Optional[torch.Tensor] = None, vec_mode: int = 0, vG: int = 0,
          vFS: int = 0, vS: int = 0, vB: int = 0, blend: Optional[torch.Tensor] = None, alpha: float = 0.0,
          out_scale: float = 1.0, out: Optional[torch.Tensor] = None, res_post: bool = False,
          out_f32: bool = False, cs_cols: int = 0, cs_scale: float = 1.0, splitk: bool = True,
          wide: bool = False, out_hw=None) -> torch.Tensor:
    """Linear layer (``geom is None``; x0 is ``[M, K]``) or convolution (``geom = (Nimg, Hin, Win)``; x0/x1 are
    channels-last with that geometry).  Returns ``[M, n_out]`` fp16.

    ``wide`` (residual-stream tensors): the result is kept as an fp16 pair - the returned tensor is ``fp16(v)`` and
    carries ``.lo = fp16(v - fp16(v))``; a ``res`` that carries ``.lo`` is added as the pair.  See ``WIDE_STREAM``.

    ``out_hw``: output extent when it is not the symmetric-padding one - taps beyond the input read zeros, so
    ``Downsample2D(padding=0)`` (``F.pad(x, (0, 1, 0, 1))`` + stride-2 conv) is ``pad = 0`` with ``out_hw = (H // 2, W // 2)``."""
    ensure_ready(x0.device)
    _need(x0, "x0")
    C0 = x0.shape[-1]
    C1 = x1.shape[-1] if x1 is not None else 0
    if x1 is not None:
        _need(x1, "x1")
    if geom is None:                      # linear layer: M one-pixel "images" (keeps pixel coordinates tiny)
        if x0.dim() != 2:
            raise RuntimeError(f"posetraj_amd.igemm: a linear layer takes a 2-D [M, K] input, got {tuple(x0.shape)}")
        Nimg, Hin, Win = x0.shape[0], 1, 1
        Hout, Wout = 1, 1
    else:
        Nimg, Hin, Win = geom
        Hs, Ws = (2 * Hin, 2 * Win) if upsample2x else (Hin, Win)
        Hout = (Hs + 2 * pw.pad_h - pw.KH) // pw.stride + 1
        Wout = (Ws + 2 * pw.pad_w - pw.KW) // pw.stride + 1
        if out_hw is not None:
            Hout, Wout = int(out_hw[0]), int(out_hw[1])
    M = Nimg * Hout * Wout
    if pw.cin != C0 + C1:
        raise RuntimeError(f"posetraj_amd.igemm: weight packed for {pw.cin} input channels, got {C0}+{C1}")
    n_out = pw.n_out
    explicit_out = out is not None
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float32 if out_f32 else torch.float16, device=x0.device)
    elif out.dtype != (torch.float32 if out_f32 else torch.float16):
        raise RuntimeError(f"posetraj_amd.igemm: `out` must be {'fp32' if out_f32 else 'fp16'}; got {out.dtype}")
    p = hip.IgemmParams()
    p.x0, p.x1 = x0.data_ptr(), _ptr(x1)
    p.C0, p.C1 = C0, C1
    p.ld0 = x0.stride(-2) if x0.dim() >= 2 else C0
    p.ld1 = (x1.stride(-2) if x1 is not None else 0)
    p.Nimg, p.Hin, p.Win, p.Hout, p.Wout = Nimg, Hin, Win, Hout, Wout
    p.KH, p.KW, p.stride, p.pad_h, p.pad_w = pw.KH, pw.KW, pw.stride, pw.pad_h, pw.pad_w
    p.upsample2x = 1 if upsample2x else 0
    p.M, p.N, p.K, p.Kpad = M, pw.N, pw.K, pw.Kpad
    p.w, p.bias = pw.w.data_ptr(), _ptr(pw.bias)
    p.out, p.ldo = out.data_ptr(), out.stride(0)
    p.res, p.ldr = _ptr(res), (res.stride(0) if res is not None else 0)
    res_lo = getattr(res, "lo", None) if res is not None else None
    if res_lo is not None and (res_lo.shape != res.shape or res_lo.stride(0) != res.stride(0) or res_lo.device != res.device
                               or res_lo.dtype != torch.float16):
        raise RuntimeError("posetraj_amd.igemm: the low half of `res` must share its shape, pitch, device and dtype")
    if res_lo is not None and getattr(res, "lo_gen", 0) != _inplace_writes.get(res.data_ptr(), 0):
        raise RuntimeError("posetraj_amd.igemm: `res` was overwritten in place after its low half was produced (stale fp16 pair); "
                           "call ops.drop_lo() on tensors whose high half is rewritten")
    p.res_lo = _ptr(res_lo)
    out_lo = None
    if wide and WIDE_STREAM and not out_f32 and not pw.geglu:
        out_lo = torch.empty_like(out)
        p.out_lo = out_lo.data_ptr()
    p.vec, p.ldv = _ptr(vec), (vec.stride(0) if vec is not None else 0)
    p.vec_mode, p.vG, p.vFS, p.vS, p.vB = (vec_mode if vec is not None else 0), vG, vFS, vS, vB
    p.blend, p.ldb, p.alpha = _ptr(blend), (blend.stride(0) if blend is not None else 0), float(alpha)
    p.out_scale = float(out_scale)
    p.act = 1 if pw.geglu else (2 if pw.silu else 0)
    p.res_post, p.out_f32 = (1 if res_post else 0), (1 if out_f32 else 0)
    p.cs_cols, p.cs_scale = int(cs_cols), float(cs_scale)
    need = hip.lib().pt_igemm_splitk_ws_bytes(C.byref(p))    # small-M layers: split-K through a caller-owned fp32 workspace
    if need > 0 and splitk:
        ws = torch.empty(need // 4, dtype=torch.float32, device=x0.device)
        p.splitk_ws, p.splitk_ws_bytes = ws.data_ptr(), need
    hip.check(hip.lib().pt_igemm_f16(C.byref(p), _stream()), "pt_igemm_f16")
    if explicit_out:
        _inplace_writes[out.data_ptr()] = _inplace_writes.get(out.data_ptr(), 0) + 1
    if out_lo is not None:
        _attach_lo(out, out_lo)
    if Profiler.shapes is not None:
        Profiler.shapes.append((M, pw.N, pw.K, pw.KH, pw.KW, pw.stride, int(upsample2x), C1, p.act,
                                int(res is not None) + 2 * int(vec is not None) + 4 * int(blend is not None) +
                                8 * int(res_lo is not None) + 16 * int(out_lo is not None)))
    return out


FUSED_FFN = os.environ.get("PT_FUSED_FFN", "1") != "0"          # (0: the two-launch form everywhere - A/B of pt_ffn_geglu_f16)
FUSED_PRE = os.environ.get("PT_FUSED_PRE", "1") != "0"          # the attention's output projection + residual + LayerNorm in that launch's prologue (0: three launches, A/B)


def ffn_fusable(w1: Packed, w2: Packed) -> bool:
    """pt_ffn_geglu_f16 serves the feed-forwards whose rows fit one workgroup: C == 320 (level 0 of the SVD nets)."""
    return (FUSED_FFN and w1.geglu and w1.K == 320 and w1.Kpad == 320 and w2.N == 320 and w2.K == w1.n_out and w2.K % 64 == 0
            and w2.KH == 1 and w2.KW == 1 and not w2.geglu and not w2.silu)


def ffn_geglu(x: torch.Tensor, w1: Packed, w2: Packed, *, res: Optional[torch.Tensor] = None, vec: Optional[torch.Tensor] = None,
              vec_mode: int = 0, vG: int = 0, vFS: int = 0, vS: int = 0, vB: int = 0, blend: Optional[torch.Tensor] = None,
              alpha: float = 0.0, out: Optional[torch.Tensor] = None, pre: Optional[dict] = None) -> torch.Tensor:
    """``w2(geglu(w1(x)))`` + the side inputs of :func:`igemm` in ONE launch (``ffn_fusable``); bit-identical to
    ``igemm(igemm(x, w1), w2, res=..., vec=..., blend=...)``.

    ``pre = dict(w=Packed, res=h, vec=, vec_mode=, vG=, vFS=, vS=, vB=, ln=(gamma, beta[, eps]))``: ``x`` is the ATTENTION OUTPUT and
    the launch also covers the three steps in front of the feed-forward - ``h = igemm(x, pre.w, res=pre.res, vec=...)``,
    ``y = layernorm(h, *ln)``, and ``+ h`` as the feed-forward's residual (``res`` must stay None) - with ``h`` kept in fp32."""
    ensure_ready(x.device)
    _need(x, "x")
    if x.dim() != 2 or x.shape[1] != w1.K:
        raise RuntimeError(f"posetraj_amd.ffn_geglu: x must be [M, {w1.K}], got {tuple(x.shape)}")
    if not ffn_fusable(w1, w2):
        raise RuntimeError("posetraj_amd.ffn_geglu: these weights are not served by pt_ffn_geglu_f16 (ops.ffn_fusable)")
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, w2.N), dtype=torch.float16, device=x.device)
    if res is not None and getattr(res, "lo", None) is not None:
        raise RuntimeError("posetraj_amd.ffn_geglu: a wide-stream residual (fp16 pair) is not supported here")
    p = hip.FfnParams()
    p.x, p.ldx = x.data_ptr(), x.stride(0)
    p.M, p.C, p.inner = M, w2.N, w2.K
    p.w1, p.b1, p.kpad1 = w1.w.data_ptr(), _ptr(w1.bias), w1.Kpad
    p.w2, p.b2, p.kpad2 = w2.w.data_ptr(), _ptr(w2.bias), w2.Kpad
    p.out, p.ldo = out.data_ptr(), out.stride(0)
    p.res, p.ldr = _ptr(res), (res.stride(0) if res is not None else 0)
    p.vec, p.ldv = _ptr(vec), (vec.stride(0) if vec is not None else 0)
    p.vec_mode, p.vG, p.vFS, p.vS, p.vB = (vec_mode if vec is not None else 0), vG, vFS, vS, vB
    p.blend, p.ldb, p.alpha = _ptr(blend), (blend.stride(0) if blend is not None else 0), float(alpha)
    if pre is not None:
        pw, pres, pv, ln = pre["w"], pre["res"], pre.get("vec"), pre["ln"]
        if res is not None:
            raise RuntimeError("posetraj_amd.ffn_geglu: with `pre` the feed-forward's residual is the projection's own output; `res` must be None")
        if pw.N != 320 or pw.K != 320 or pw.KH * pw.KW != 1 or pw.geglu or pw.silu or getattr(pres, "lo", None) is not None:
            raise RuntimeError("posetraj_amd.ffn_geglu: `pre.w` must be a plain 320 -> 320 linear pack and `pre.res` a plain fp16 tensor")
        _need(pres, "pre.res")
        p.pre_w, p.pre_b, p.pre_kpad = pw.w.data_ptr(), _ptr(pw.bias), pw.Kpad
        p.pre_res, p.pre_ldr = pres.data_ptr(), pres.stride(0)
        if pv is not None:
            p.pre_vec, p.pre_ldv, p.pre_vec_mode = pv.data_ptr(), pv.stride(0), int(pre.get("vec_mode", 1))
            p.pre_vG, p.pre_vFS, p.pre_vS, p.pre_vB = int(pre.get("vG", 0)), int(pre.get("vFS", 0)), int(pre.get("vS", 0)), int(pre.get("vB", 0))
        p.ln_gamma, p.ln_beta, p.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2]) if len(ln) > 2 else 1e-5
    hip.check(hip.lib().pt_ffn_geglu_f16(C.byref(p), _stream()), "pt_ffn_geglu_f16")
    if Profiler.shapes is not None:
        Profiler.shapes.append((M, w2.N, w2.K, 1, 1, 1, 0, 0, 3, int(res is not None) + 2 * int(vec is not None) + 4 * int(blend is not None)
                                + 32 * int(pre is not None)))              # 32: with the out-projection + LayerNorm prologue (its residual is a read too)
    return out


FUSED_LNLIN = os.environ.get("PT_FUSED_LNLIN", "1") != "0"      # LayerNorm + Q/K/V projection of the 320-channel level in one launch (0: two launches, A/B)


def ln_linear_fusable(x: torch.Tensor, pw: Packed) -> bool:
    """pt_ln_linear_f16 serves LayerNorm -> bias-free linear layer where a workgroup holds whole rows in registers: K == 320."""
    return (FUSED_LNLIN and pw.K == 320 and pw.Kpad == 320 and pw.bias is None and pw.KH == 1 and pw.KW == 1 and not pw.geglu
            and not pw.silu and pw.N % 8 == 0 and x.dim() == 2 and x.shape[1] == 320 and getattr(x, "lo", None) is None)


def ln_linear(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, pw: Packed, eps: float = 1e-5, *, cs_cols: int = 0,
              cs_scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``igemm(layernorm(x, gamma, beta, eps), pw, cs_cols=, cs_scale=)`` in ONE launch (``ln_linear_fusable``): the normalised rows
    stay in registers.  Equal to the two launches up to the summation order of the row statistics."""
    ensure_ready(x.device)
    _need(x, "x")
    if not ln_linear_fusable(x, pw):
        raise RuntimeError("posetraj_amd.ln_linear: this layer is not served by pt_ln_linear_f16 (ops.ln_linear_fusable)")
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, pw.N), dtype=torch.float16, device=x.device)
    p = hip.LnLinParams()
    p.x, p.ldx = x.data_ptr(), x.stride(0)
    p.M, p.N, p.K = M, pw.N, pw.K
    p.w, p.kpad, p.bias = pw.w.data_ptr(), pw.Kpad, None
    p.ln_gamma, p.ln_beta, p.ln_eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
    p.out, p.ldo = out.data_ptr(), out.stride(0)
    p.cs_cols, p.cs_scale = int(cs_cols), float(cs_scale)
    hip.check(hip.lib().pt_ln_linear_f16(C.byref(p), _stream()), "pt_ln_linear_f16")
    if Profiler.shapes is not None:
        Profiler.shapes.append((M, pw.N, pw.K, 1, 1, 1, 0, 0, 4, 0))        # act 4: LayerNorm in the prologue
    return out


def groupnorm(x0: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, rows_per_sample: int, n_samples: int,
              eps: float, silu: bool, x1: Optional[torch.Tensor] = None, groups: int = 32) -> torch.Tensor:
    """GroupNorm (+SiLU) of channels-last data viewed as ``[rows, C]``; two sources are emitted concatenated."""
    ensure_ready(x0.device)
    _need(x0, "x0")
    C0 = x0.shape[-1]
    C1 = x1.shape[-1] if x1 is not None else 0
    Ct = C0 + C1
    rows = rows_per_sample * n_samples
    L = hip.lib()
    nfl = L.pt_groupnorm_scratch_floats(rows, Ct, n_samples)
    partials = torch.empty(nfl, dtype=torch.float32, device=x0.device)
    st = _stream()
    hip.check(L.pt_groupnorm_stats(x0.data_ptr(), _ptr(x1), C0, C1, groups, rows_per_sample, n_samples, partials.data_ptr(), st),
              "pt_groupnorm_stats")
    y = torch.empty((rows, Ct), dtype=torch.float16, device=x0.device)
    hip.check(L.pt_groupnorm_apply(x0.data_ptr(), _ptr(x1), C0, C1, groups, rows_per_sample, n_samples, float(eps),
                                   gamma.data_ptr(), beta.data_ptr(), partials.data_ptr(), 1 if silu else 0, y.data_ptr(), st),
              "pt_groupnorm_apply")
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, *,
              vec: Optional[torch.Tensor] = None, vG: int = 0) -> torch.Tensor:
    _need(x, "x")
    M, Cc = x.shape
    y = torch.empty_like(x)
    hip.check(hip.lib().pt_layernorm_f16(x.data_ptr(), M, Cc, _ptr(vec), (vec.stride(0) if vec is not None else 0),
                                         1 if vec is not None else 0, vG, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                         y.data_ptr(), _stream()), "pt_layernorm_f16")
    return y


def attn_spatial(qkv: torch.Tensor, Nimg: int, S: int, heads: int, head_dim: int, q_prescaled: bool = False) -> torch.Tensor:
    """``q_prescaled``: the Q third of ``qkv`` was produced with ``cs_scale = attn_q_prescale(head_dim)``."""
    ensure_ready(qkv.device)
    _need(qkv, "qkv")
    Cc = heads * head_dim
    out = torch.empty((Nimg * S, Cc), dtype=torch.float16, device=qkv.device)
    hip.check(hip.lib().pt_attn_spatial_f16(qkv.data_ptr(), qkv.stride(0), Cc, 2 * Cc, out.data_ptr(), Cc, Nimg, S,
                                            heads, head_dim, head_dim ** -0.5, 1 if q_prescaled else 0, _stream()),
              "pt_attn_spatial_f16")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nbatch: int, Sq: int, Sk: int, heads: int,
              head_dim: int) -> torch.Tensor:
    """General flash attention (``pt_attn_f16``): ``q [nbatch*Sq, >= heads*head_dim]``, ``k`` / ``v`` ``[nbatch*Sk, ...]`` - 2-D
    fp16 tensors or column-block views of one fused projection (row pitch = ``stride(0)``)."""
    ensure_ready(q.device)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, n)
        if t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"posetraj_amd.attention: `{n}` must be a 2-D tensor with unit column stride")
    Cc = heads * head_dim
    if q.shape != (nbatch * Sq, Cc) or k.shape != (nbatch * Sk, Cc) or v.shape != (nbatch * Sk, Cc):
        raise RuntimeError(f"posetraj_amd.attention: shapes {tuple(q.shape)} / {tuple(k.shape)} / {tuple(v.shape)} for "
                           f"{nbatch} x ({Sq}, {Sk}) tokens of {heads} x {head_dim}")
    out = torch.empty((nbatch * Sq, Cc), dtype=torch.float16, device=q.device)
    hip.check(hip.lib().pt_attn_f16(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                    out.data_ptr(), Cc, nbatch, Sq, Sk, heads, head_dim, head_dim ** -0.5, _stream()), "pt_attn_f16")
    return out


def attn_q_prescale(head_dim: int) -> float:
    """softmax scale * log2(e): what the spatial-attention kernel wants folded into Q (``cs_scale`` of the QKV GEMM)."""
    return head_dim ** -0.5 * 1.4426950408889634


def attn_temporal(qkv: torch.Tensor, B: int, F: int, S: int, heads: int, head_dim: int) -> torch.Tensor:
    _need(qkv, "qkv")
    Cc = heads * head_dim
    out = torch.empty((B * F * S, Cc), dtype=torch.float16, device=qkv.device)
    hip.check(hip.lib().pt_attn_temporal_f16(qkv.data_ptr(), qkv.stride(0), Cc, 2 * Cc, out.data_ptr(), Cc, B, F, S,
                                             heads, head_dim, head_dim ** -0.5, _stream()), "pt_attn_temporal_f16")
    return out


def axpy(a: torch.Tensor, r: torch.Tensor, m: float) -> torch.Tensor:
    _need(a, "a"); _need(r, "r")
    if a.numel() != r.numel():
        raise RuntimeError(f"posetraj_amd.axpy: size mismatch {tuple(a.shape)} vs {tuple(r.shape)}")
    out = torch.empty_like(a)
    hip.check(hip.lib().pt_axpy_f16(a.data_ptr(), r.data_ptr(), float(m), out.data_ptr(), a.numel(), _stream()),
              "pt_axpy_f16")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _need(x, "x")
    y = torch.empty_like(x)
    hip.check(hip.lib().pt_silu_f16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "pt_silu_f16")
    return y


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    _need(t, "t", torch.float32)
    n = t.numel()
    out = torch.empty((n, dim), dtype=torch.float16, device=t.device)
    hip.check(hip.lib().pt_timestep_embedding(t.data_ptr(), n, dim, out.data_ptr(), _stream()), "pt_timestep_embedding")
    return out


def to_channels_last(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    """``[N, C, H, W]`` (fp16/fp32, any strides) -> contiguous fp16 ``[N, H, W, Cpad]``.  Zero-copy when ``x`` already
    is a permuted view of such a buffer."""
    if not x.is_cuda:
        raise RuntimeError("posetraj_amd: inputs must live on the GPU (no CPU path exists)")
    N, Cc, H, W = x.shape
    cpad = cpad or Cc
    v = x.permute(0, 2, 3, 1)
    if x.dtype == torch.float16 and cpad == Cc and v.is_contiguous():
        return v
    if x.dtype not in (torch.float16, torch.float32):
        x = x.float()
    x = x.contiguous()
    out = torch.empty((N, H, W, cpad), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().pt_nchw_to_nhwc_f16(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, N, Cc, H, W, cpad,
                                            out.data_ptr(), _stream()), "pt_nchw_to_nhwc_f16")
    return out


def to_nchw(x: torch.Tensor, Cc: Optional[int] = None, f32: bool = False) -> torch.Tensor:
    """channels-last fp16 ``[N, H, W, ld]`` -> contiguous ``[N, C, H, W]``."""
    _need(x, "x")
    N, H, W, ld = x.shape
    Cc = Cc or ld
    out = torch.empty((N, Cc, H, W), dtype=torch.float32 if f32 else torch.float16, device=x.device)
    hip.check(hip.lib().pt_nhwc_to_nchw(x.data_ptr(), N, Cc, H, W, ld, out.data_ptr(), 1 if f32 else 0, _stream()),
              "pt_nhwc_to_nchw")
    return out


def concat_camera(feat: torch.Tensor, cam: torch.Tensor, cpad: int) -> torch.Tensor:
    _need(feat, "feat"); _need(cam, "cam")
    N, H, W, Cc = feat.shape
    out = torch.empty((N, H, W, cpad), dtype=torch.float16, device=feat.device)
    hip.check(hip.lib().pt_concat_camera(feat.data_ptr(), Cc, cam.data_ptr(), N, H * W, cpad, out.data_ptr(), _stream()),
              "pt_concat_camera")
    return out


def scale_concat_input(latents: torch.Tensor, image_latents: torch.Tensor, sigma: float,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 ``latents [Bc, F, 4, h, w]`` + fp16 ``image_latents [2Bc, 4, h, w]`` -> fp16 ``[2Bc, F, h, w, 8]``."""
    _need(latents, "latents", torch.float32); _need(image_latents, "image_latents")
    Bc, F, _, h, w = latents.shape
    if out is None:
        out = torch.empty((2 * Bc, F, h, w, 8), dtype=torch.float16, device=latents.device)
    hip.check(hip.lib().pt_scale_concat_input(latents.data_ptr(), image_latents.data_ptr(), float(sigma), Bc, F, h, w,
                                              out.data_ptr(), _stream()), "pt_scale_concat_input")
    return out


def cfg_euler_step(noise_pred: torch.Tensor, guidance: torch.Tensor, sigma: float, sigma_next: float,
                   prediction_type: int, latents: torch.Tensor) -> None:
    """In place on fp32 ``latents [Bc, F, 4, h, w]``; ``noise_pred`` fp16 or fp32 channels-last ``[2Bc, F, h, w, ld]``."""
    f32 = noise_pred.dtype == torch.float32
    _need(noise_pred, "noise_pred", noise_pred.dtype if f32 else torch.float16)
    _need(latents, "latents", torch.float32); _need(guidance, "guidance", torch.float32)
    Bc, F, _, h, w = latents.shape
    hip.check(hip.lib().pt_cfg_euler_step(noise_pred.data_ptr(), 1 if f32 else 0, noise_pred.stride(-2), guidance.data_ptr(), float(sigma),
                                          float(sigma_next), prediction_type, Bc, F, h, w, latents.data_ptr(), _stream()),
              "pt_cfg_euler_step")


def scale(x: torch.Tensor, k: float) -> torch.Tensor:
    if not x.is_cuda or x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("posetraj_amd.scale: needs a fp16/fp32 tensor on the GPU (no CPU path exists)")
    x = x.contiguous()
    y = torch.empty_like(x)
    hip.check(hip.lib().pt_scale(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, float(k), y.data_ptr(), x.numel(),
                                 _stream()), "pt_scale")
    return y


def euler_step(model_output: torch.Tensor, sample_f32: torch.Tensor, sigma: float, sigma_next: float,
               prediction_type: int) -> torch.Tensor:
    if not model_output.is_cuda or model_output.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("posetraj_amd.euler_step: needs a fp16/fp32 model_output on the GPU (no CPU path exists)")
    _need(sample_f32, "sample", torch.float32)
    mo = model_output.contiguous()
    x = sample_f32.contiguous()
    out = torch.empty_like(x)
    hip.check(hip.lib().pt_euler_step(mo.data_ptr(), 1 if mo.dtype == torch.float32 else 0, x.data_ptr(), float(sigma),
                                      float(sigma_next), prediction_type, out.data_ptr(), x.numel(), _stream()),
              "pt_euler_step")
    return out


def add_noise(x: torch.Tensor, noise: torch.Tensor, sigma_per_sample: torch.Tensor) -> torch.Tensor:
    """``x + noise * sigma[b]`` in the dtype of ``x`` (fp16 / fp32); ``sigma_per_sample`` fp32 ``[batch]`` on the device."""
    if not x.is_cuda or x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("posetraj_amd.add_noise: needs fp16/fp32 tensors on the GPU (no CPU path exists)")
    _need(noise, "noise", x.dtype); _need(sigma_per_sample, "sigma_per_sample", torch.float32)
    if noise.shape != x.shape or sigma_per_sample.numel() != x.shape[0]:
        raise RuntimeError(f"posetraj_amd.add_noise: shapes {tuple(x.shape)} / {tuple(noise.shape)} / {tuple(sigma_per_sample.shape)}")
    xc, nc = x.contiguous(), noise.contiguous()
    y = torch.empty_like(xc)
    hip.check(hip.lib().pt_add_noise(xc.data_ptr(), nc.data_ptr(), 1 if x.dtype == torch.float32 else 0,
                                     sigma_per_sample.contiguous().data_ptr(), xc.numel() // x.shape[0], y.data_ptr(),
                                     xc.numel(), _stream()), "pt_add_noise")
    return y


def resize_with_antialiasing(image: torch.Tensor, size) -> torch.Tensor:
    """``_resize_with_antialiasing`` of the reference pipeline (``pipeline...:604-634``): ``[B, C, H, W]`` (or ``[C, H, W]``)
    fp32 on the GPU -> ``[B, C, size[0], size[1]]`` fp32.  Sigma / kernel size / taps follow the reference formulas on the
    host (a handful of floats); blur and bicubic interpolation run in ``pt_resize_antialias_f32``."""
    _need(image, "image", torch.float32)
    if image.dim() == 3:
        image = image.unsqueeze(0)
    B, Cc, H, W = image.shape
    oh, ow = int(size[0]), int(size[1])
    factors = (H / oh, W / ow)
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]

    def taps(window, sigma):                                  # _gaussian (:676-689), fp32 like the reference
        x = torch.arange(window, dtype=torch.float32) - window // 2
        g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
        return (g / g.sum()).to(image.device)

    tx, ty = taps(ks[1], sigmas[1]), taps(ks[0], sigmas[0])
    x = image.contiguous()
    tmp = torch.empty(2 * x.numel(), dtype=torch.float32, device=x.device)
    out = torch.empty((B, Cc, oh, ow), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().pt_resize_antialias_f32(x.data_ptr(), B * Cc, H, W, oh, ow, tx.data_ptr(), ks[1], ty.data_ptr(), ks[0],
                                                tmp.data_ptr(), out.data_ptr(), _stream()), "pt_resize_antialias_f32")
    return out


def patchify(image: torch.Tensor, patch: int, ld: int) -> torch.Tensor:
    """``[B, C, H, W]`` fp32 / fp16 -> fp16 ``[B * (H/P) * (W/P), ld]`` rows of flattened patches in (c, ky, kx) order."""
    if not image.is_cuda or image.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("posetraj_amd.patchify: needs a fp16/fp32 image on the GPU (no CPU path exists)")
    x = image.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B * (H // patch) * (W // patch), ld), dtype=torch.float16, device=x.device)
    hip.check(hip.lib().pt_patchify_f16(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, B, Cc, H, W, patch, ld, out.data_ptr(),
                                        _stream()), "pt_patchify_f16")
    return out


def activation(x: torch.Tensor, kind: str) -> torch.Tensor:
    """``gelu`` (erf) / ``quick_gelu`` on an fp16 tensor."""
    _need(x, "x")
    k = {"gelu": 1, "quick_gelu": 2}.get(kind)
    if k is None:
        raise ValueError(f"posetraj_amd.activation: hidden_act {kind!r} unsupported (gelu, quick_gelu)")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    hip.check(hip.lib().pt_act_f16(xc.data_ptr(), y.data_ptr(), xc.numel(), k, _stream()), "pt_act_f16")
    return y


def vae_time_conv_out(x: torch.Tensor, w_host, b_host, F: int, HW: int, out: torch.Tensor) -> torch.Tensor:
    """``time_conv_out`` of one ``vae.decode`` call: ``x`` fp32 channels-last ``[F*HW, ld]`` -> ``out`` fp32 ``[F, 3, ...]`` (a
    contiguous slice of the caller's frame buffer).  ``w_host`` / ``b_host``: ctypes float arrays (27 / 3 values)."""
    _need(x, "x", torch.float32); _need(out, "out", torch.float32)
    if x.dim() != 2 or x.shape[0] != F * HW or out.numel() != F * 3 * HW or not out.is_contiguous():
        raise RuntimeError(f"posetraj_amd.vae_time_conv_out: shapes {tuple(x.shape)} / {tuple(out.shape)} for {F} x {HW}")
    hip.check(hip.lib().pt_vae_time_conv_out(x.data_ptr(), x.stride(0), w_host, b_host, F, HW, out.data_ptr(), _stream()),
              "pt_vae_time_conv_out")
    return out


def frames_postprocess(clip: torch.Tensor, output_type: str) -> torch.Tensor:
    """``tensor2vid`` for one clip (``pipeline...:70-83`` + ``VaeImageProcessor.postprocess``): fp32 ``[F, 3, H, W]`` ->
    "pt": fp32 ``[F, 3, H, W]`` in [0, 1]; "np": fp32 ``[F, H, W, 3]``; "pil": uint8 ``[F, H, W, 3]``."""
    _need(clip, "clip", torch.float32)
    mode = {"pt": 0, "np": 1, "pil": 2}[output_type]
    F, Cc, H, W = clip.shape
    if Cc != 3:
        raise RuntimeError(f"posetraj_amd.frames_postprocess: 3-channel frames expected, got {Cc}")
    x = clip.contiguous()
    shape = (F, 3, H, W) if mode == 0 else (F, H, W, 3)
    out = torch.empty(shape, dtype=torch.uint8 if mode == 2 else torch.float32, device=x.device)
    hip.check(hip.lib().pt_frames_postprocess(x.data_ptr(), F, H * W, mode, out.data_ptr(), _stream()), "pt_frames_postprocess")
    return out


def to_nchw_f32(x: torch.Tensor, N: int, HW: int, Cc: int) -> torch.Tensor:
    """fp32 channels-last ``[N*HW, ld]`` -> fp32 ``[N, C, HW]``."""
    _need(x, "x", torch.float32)
    out = torch.empty((N, Cc, HW), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().pt_nhwc_to_nchw_f32(x.data_ptr(), N, Cc, HW, x.stride(0), out.data_ptr(), _stream()), "pt_nhwc_to_nchw_f32")
    return out


def gaussian_sample(params: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """``DiagonalGaussianDistribution.sample``: ``params`` fp32 ``[N, 2C, h, w]`` (mean | logvar), ``noise`` fp32 ``[N, C, h, w]``."""
    _need(params, "params", torch.float32); _need(noise, "noise", torch.float32)
    N, C2, h, w = params.shape
    out = torch.empty((N, C2 // 2, h, w), dtype=torch.float32, device=params.device)
    hip.check(hip.lib().pt_gaussian_sample(params.contiguous().data_ptr(), noise.contiguous().data_ptr(), N, C2 // 2, h * w,
                                           out.data_ptr(), _stream()), "pt_gaussian_sample")
    return out


class Profiler:
    """hipEvent bracketing of every igemm / spatial-attention launch (bench.py's roofline leg)."""
    FAMILIES = {"igemm": 0, "attn_spatial": 1}
    shapes = None          # when a list: ops.igemm appends one shape tuple per launch (tools/shape_report.py)

    def __enter__(self):
        hip.check(hip.lib().pt_prof_enable(1), "pt_prof_enable")
        return self

    def __exit__(self, *exc):
        hip.check(hip.lib().pt_prof_enable(0), "pt_prof_enable")

    @staticmethod
    def collect_list(family: str, cap: int = 1 << 20):
        ms, fl = (C.c_double * cap)(), (C.c_double * cap)()
        n = hip.lib().pt_prof_collect_list(Profiler.FAMILIES[family], ms, fl, cap)
        if n < 0:
            raise RuntimeError("pt_prof_collect_list failed")
        return list(ms[:n]), list(fl[:n])

    @staticmethod
    def collect(family: str):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        hip.check(hip.lib().pt_prof_collect(Profiler.FAMILIES[family], C.byref(n), C.byref(ms), C.byref(fl)),
                  "pt_prof_collect")
        return dict(launches=n.value, ms=ms.value, flops=fl.value)
