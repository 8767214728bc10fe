"""Kernel-level parity on the MI355X: every HIP entry point, called through the C ABI (posetraj_amd.ops -> ctypes),
against a plain PyTorch fp32 computation of the same op on the same fp16-rounded inputs.
Tolerance: rel-L2 <= 4e-4 for every kernel whose error is its fp16 output rounding (2.07e-4 for uniformly distributed
mantissas; measured 2.0 - 2.4e-4 across the sweep: 1.7 x margin), 8e-4 for the attention kernels (fp16 P operand: measured <= 4.5e-4).
VERDICT r02 #9: the sweep ran at 2e-3, ten times the measured error."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 4e-4
TOL_ATTN = 8e-4


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from posetraj_amd import ops
    return ops


def h16(*shape, g, scale=1.0, dev):
    return (torch.randn(*shape, generator=g) * scale).half().to(dev)


# ------------------------------------------------------------------------------------------------- linear
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (1000, 960, 320), (77, 64, 1024), (2, 1280, 320),
                                   (4032, 1280, 1280), (513, 2560, 640), (80640, 64, 64), (258048, 320, 320)])
def test_linear_bias(ops, dev, M, N, K):
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    y = ops.igemm(x, pack_linear(w, b, dev))
    ref = F.linear(x.float(), w.float(), b.float())
    assert y.shape == (M, N)
    assert rel(y, ref) < TOL


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(700, 320, 320), (1030, 960, 640), (513, 2560, 320), (300, 1280, 1280), (2, 64, 128),
                                   (260, 192, 64), (260, 192, 32 * 3)])
def test_linear_every_tile_config(ops, dev, cfg, M, N, K):
    """Each tile configuration of the implicit-GEMM kernel on ragged M / N, full epilogue."""
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + N + K + cfg)
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    res = h16(M, N, g=g, dev=dev)
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        y = ops.igemm(x, pack_linear(w, b, dev), res=res, out_scale=0.5)
        ref = 0.5 * (F.linear(x.float(), w.float(), b.float()) + res.float())
        assert rel(y, ref) < TOL
        if cfg not in (1, 5) and N % 32 == 0:                   # 128x320 and 256x32 have no GEGLU pairing
            yg = ops.igemm(x, pack_linear(w, b, dev, geglu=True))
            hh, gg = F.linear(x.float(), w.float(), b.float()).chunk(2, dim=-1)
            assert rel(yg, hh * F.gelu(gg)) < TOL
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_conv_every_tile_config(ops, dev, cfg):
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(40 + cfg)
    N, H, W, C0, C1, Co = 2, 11, 13, 128, 64, 320
    a, s = h16(N, H, W, C0, g=g, dev=dev), h16(N, H, W, C1, g=g, dev=dev)
    w, b = h16(Co, C0 + C1, 3, 3, g=g, scale=(9 * (C0 + C1)) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        y = ops.igemm(a, pack_conv2d(w, b, dev), x1=s, geom=(N, H, W))
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))
    cat = torch.cat([a, s], dim=-1).float().permute(0, 3, 1, 2)
    ref = F.conv2d(cat, w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert rel(y.view(ref.shape), ref) < TOL


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_conv_variants_every_tile_config(ops, dev, cfg):
    """stride 2, nearest-2x upsampling, SiLU and the full row-wise tail (residual + row vector + blend + scale) under
    each tile configuration - the pipelined kernels share the gather / epilogue code but not the staging order."""
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(70 + cfg)
    N, H, W, Ci, Co = 3, 10, 14, 128, 320
    x = h16(N, H, W, Ci, g=g, dev=dev)
    w, b = h16(Co, Ci, 3, 3, g=g, scale=(9 * Ci) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
    xc = x.float().permute(0, 3, 1, 2)
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        y2 = ops.igemm(x, pack_conv2d(w, b, dev, stride=2), geom=(N, H, W))
        yu = ops.igemm(x, pack_conv2d(w, b, dev), geom=(N, H, W), upsample2x=True)
        res, blend = h16(N * H * W, Co, g=g, dev=dev), h16(N * H * W, Co, g=g, dev=dev)
        vec = h16(N, Co, g=g, dev=dev)
        yt = ops.igemm(x, pack_conv2d(w, b, dev), geom=(N, H, W), res=res, vec=vec, vec_mode=1, vG=H * W, blend=blend,
                       alpha=0.3, out_scale=0.5)
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))
    r2 = F.conv2d(xc, w.float(), b.float(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert rel(y2.view(r2.shape), r2) < TOL
    ru = F.conv2d(F.interpolate(xc, scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert rel(yu.view(ru.shape), ru) < TOL
    rt = F.conv2d(xc, w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(N * H * W, Co)
    rt = rt + res.float() + vec.float().repeat_interleave(H * W, dim=0)
    rt = 0.5 * (0.3 * blend.float() + 0.7 * rt)
    assert rel(yt, rt) < TOL


@pytest.mark.parametrize("Ci,Co,stride,silu,f32", [(8, 16, 1, True, False), (16, 32, 2, True, False), (32, 96, 2, False, False), (320, 4, 1, False, True),
                                                    (128, 8, 1, False, False), (512, 3, 1, False, False)])
def test_narrow_output_convolutions_take_the_256x32_tiles(ops, dev, Ci, Co, stride, silu, f32):
    """N <= 96 (the condition encoder's 16 / 32 / 96 channels, conv_out's 4, the VAE's 8 and 3): the automatic choice is the
    256 x 32 configuration; SiLU epilogue, stride 2, fp32 output and odd N all go through it."""
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(Ci + Co)
    N, H, W = 2, 19, 23
    x = h16(N, H, W, Ci, g=g, dev=dev)
    w, b = h16(Co, Ci, 3, 3, g=g, scale=(9 * Ci) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
    pw = pack_conv2d(w, b, dev, stride=stride)
    pw.silu = silu
    y = ops.igemm(x, pw, geom=(N, H, W), out_f32=f32)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), stride=stride, padding=1)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
    assert y.dtype == (torch.float32 if f32 else torch.float16) and rel(y.view(ref.shape), ref) < TOL


@pytest.mark.parametrize("cfg", [0, 3])
@pytest.mark.parametrize("M,N,K,geglu", [(16128, 1280, 1280, False), (4096, 2560, 320, True), (1000, 640, 5760, False)])
def test_pipelined_kernels_race_screen(ops, dev, cfg, M, N, K, geglu):
    """The ping-pong main loops order LDS-DMA landings against fragment reads by counted vmcnt + barriers only; an
    early read would pass a single check whenever the DMA happened to land first.  Screen: many launches over a full
    chip of tiles must be bitwise identical and match the fp32 reference."""
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + N + K + cfg)
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    pw = pack_linear(w, b, dev, geglu=geglu)
    res = h16(M, pw.n_out, g=g, dev=dev)
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        first = ops.igemm(x, pw, res=res).clone()
        out = torch.empty_like(first)
        for _ in range(60):
            ops.igemm(x, pw, res=res, out=out)
            assert torch.equal(out, first)
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))
    lin = F.linear(x.float(), w.float(), b.float())
    if geglu:
        hh, gg = lin.chunk(2, dim=-1)
        lin = hh * F.gelu(gg)
    assert rel(first, lin + res.float()) < TOL


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("vG", [512, 300, 1024])
def test_row_vector_with_tile_aligned_and_straddling_periods(ops, dev, cfg, vG):
    """A broadcast row vector (side input of the tail, one row index per output row): periods that cover whole tiles
    (vG = 512, 1024) and one that straddles tiles and chunks (vG = 300).  All must give bias + x W^T + res + vec[m // vG]."""
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_linear
    M, N, K = 1024, 320, 192
    g = torch.Generator().manual_seed(vG + cfg)
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    res = h16(M, N, g=g, dev=dev)
    nv = -(-M // vG)
    vec = h16(nv, N, g=g, dev=dev)
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        y = ops.igemm(x, pack_linear(w, b, dev), res=res, vec=vec, vec_mode=1, vG=vG)
        y2 = ops.igemm(x, pack_linear(w, b, dev), vec=vec, vec_mode=1, vG=vG)
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))
    lin = F.linear(x.float(), w.float(), b.float())
    vv = vec.float()[torch.arange(M, device=dev) // vG]
    assert rel(y, lin + res.float() + vv) < TOL
    assert rel(y2, lin + vv) < TOL


def test_linear_a_equals_identity_asymmetric_b(ops, dev):
    """A = I with an asymmetric B catches a transposed C write (cdna guide, 3)."""
    from posetraj_amd.packing import pack_linear
    K = 128
    x = torch.eye(K).half().to(dev)
    w = (torch.arange(K * K).reshape(K, K) % 251 - 125).float().div(64).half().to(dev)       # exact in fp16
    y = ops.igemm(x, pack_linear(w, None, dev))
    assert torch.equal(y.float().cpu(), w.float().cpu().t())


@pytest.mark.parametrize("vec_mode", [1, 2])
def test_linear_full_epilogue(ops, dev, vec_mode):
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(9 + vec_mode)
    B, Fr, S, C, K = 2, 3, 50, 192, 256
    M = B * Fr * S
    x, w, b = h16(M, K, g=g, dev=dev), h16(C, K, g=g, scale=K ** -0.5, dev=dev), h16(C, g=g, dev=dev)
    res, blend = h16(M, C, g=g, dev=dev), h16(M, C, g=g, dev=dev)
    vec = h16(B, 3 * C, g=g, dev=dev)
    vsl = vec[:, C:2 * C]                                   # strided slice, like the stacked per-forward tables
    alpha, scale = 0.37, 0.8
    kw = dict(vec_mode=1, vG=Fr * S) if vec_mode == 1 else dict(vec_mode=2, vFS=Fr * S, vS=S, vB=B)
    y = ops.igemm(x, pack_linear(w, b, dev), res=res, vec=vsl, blend=blend, alpha=alpha, out_scale=scale, **kw)
    m = torch.arange(M, device=dev)
    idx = m // (Fr * S) if vec_mode == 1 else ((m // (Fr * S)) * S + m % S) % B
    t = F.linear(x.float(), w.float(), b.float()) + res.float() + vsl.float()[idx]
    ref = scale * (alpha * blend.float() + (1 - alpha) * t)
    assert rel(y, ref) < TOL


def test_linear_geglu(ops, dev):
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(3)
    M, C = 333, 320
    x, w, b = h16(M, C, g=g, dev=dev), h16(8 * C, C, g=g, scale=C ** -0.5, dev=dev), h16(8 * C, g=g, dev=dev)
    y = ops.igemm(x, pack_linear(w, b, dev, geglu=True))
    hh, gg = F.linear(x.float(), w.float(), b.float()).chunk(2, dim=-1)
    assert y.shape == (M, 4 * C)
    assert rel(y, hh * F.gelu(gg)) < TOL


# ------------------------------------------------------------------------------------------------- fused GEGLU feed-forward (C = 320)
@pytest.mark.parametrize("M,side", [(128, "res"), (1000, "res"), (4096, "res+vec"), (2304, "res+blend"), (70, "none"), (33000, "res")])
def test_fused_feed_forward_equals_the_two_launch_form(ops, dev, M, side):
    """pt_ffn_geglu_f16 (one launch, the [M, 1280] intermediate stays on the CU) against igemm(GEGLU) -> igemm(+ side inputs): the
    same packed weights, the same fp16 rounding of the intermediate, the same fp32 accumulation order - bit for bit - and against
    the fp32 computation of diffusers' FeedForward(geglu) + residual.  Ragged M (1000, 70) exercises the clamped rows."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M)
    C, I = 320, 1280
    x = h16(M, C, g=g, dev=dev)
    w1, b1 = h16(2 * I, C, g=g, scale=C ** -0.5, dev=dev), h16(2 * I, g=g, scale=0.3, dev=dev)
    w2, b2 = h16(C, I, g=g, scale=I ** -0.5, dev=dev), h16(C, g=g, scale=0.3, dev=dev)
    p1, p2 = pack_linear(w1, b1, dev, geglu=True), pack_linear(w2, b2, dev)
    assert ops.ffn_fusable(p1, p2)
    res = h16(M, C, g=g, dev=dev) if "res" in side else None
    kw = {}
    S = 64
    if "vec" in side:
        kw = dict(vec=h16(M // S, C, g=g, dev=dev), vec_mode=1, vG=S)
    if "blend" in side:
        kw = dict(blend=h16(M, C, g=g, dev=dev), alpha=0.37)
    two = ops.igemm(ops.igemm(x, p1), p2, res=res, **kw)
    one = ops.ffn_geglu(x, p1, p2, res=res, **kw)
    torch.cuda.synchronize()
    assert one.shape == two.shape == (M, C)
    assert torch.equal(one, two), f"{int((one != two).sum())} of {one.numel()} values differ, max |d| {float((one.float() - two.float()).abs().max())}"
    hh, gg = F.linear(x.float(), w1.float(), b1.float()).chunk(2, dim=-1)
    ref = F.linear((hh * F.gelu(gg)).half().float(), w2.float(), b2.float())
    if res is not None:
        ref = ref + res.float()
    if "vec" in side:
        ref = ref + kw["vec"].float().repeat_interleave(S, dim=0)
    if "blend" in side:
        ref = 0.37 * kw["blend"].float() + (1 - 0.37) * ref
    assert rel(one, ref) < TOL


@pytest.mark.parametrize("M,mode", [(256, 1), (1000, 1), (4608, 2), (70, 2)])
def test_fused_feed_forward_with_projection_and_layernorm(ops, dev, M, mode):
    """``pre=``: the launch starts at the attention output - out-projection + residual + cross-attention row vector (both index
    modes), LayerNorm, feed-forward + residual (+ AlphaBlender in mode 2) - against the four-launch composition it replaces and the
    fp32 computation.  h stays fp32 inside the kernel where the composition rounds it to fp16: the fused result may only be CLOSER."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + mode)
    C, I, S, Fr = 320, 1280, 64, 2
    a, h0 = h16(M, C, g=g, dev=dev), h16(M, C, g=g, scale=1.5, dev=dev)
    wo, bo = h16(C, C, g=g, scale=C ** -0.5, dev=dev), h16(C, g=g, scale=0.2, dev=dev)
    w1, b1 = h16(2 * I, C, g=g, scale=C ** -0.5, dev=dev), h16(2 * I, g=g, scale=0.3, dev=dev)
    w2, b2 = h16(C, I, g=g, scale=I ** -0.5, dev=dev), h16(C, g=g, scale=0.3, dev=dev)
    gam, bet = (1.0 + 0.2 * torch.randn(C, generator=g)).half().to(dev), (0.1 * torch.randn(C, generator=g)).half().to(dev)
    po, p1, p2 = pack_linear(wo, bo, dev), pack_linear(w1, b1, dev, geglu=True), pack_linear(w2, b2, dev)
    if mode == 1:
        vec = h16((M + Fr * S - 1) // (Fr * S), C, g=g, dev=dev)
        vkw = dict(vec=vec, vec_mode=1, vG=Fr * S)
        vidx = torch.arange(M) // (Fr * S)
    else:
        vB = 2
        vec = h16(vB, C, g=g, dev=dev)
        vkw = dict(vec=vec, vec_mode=2, vFS=Fr * S, vS=S, vB=vB)
        m = torch.arange(M)
        vidx = ((m // (Fr * S)) * S + m % S) % vB
    blend = h16(M, C, g=g, dev=dev) if mode == 2 else None
    bkw = dict(blend=blend, alpha=0.37) if mode == 2 else {}
    h = ops.igemm(a, po, res=h0, **vkw)
    comp = ops.igemm(ops.igemm(ops.layernorm(h, gam, bet), p1), p2, res=h, **bkw)
    one = ops.ffn_geglu(a, p1, p2, pre=dict(w=po, res=h0, ln=(gam, bet, 1e-5), **vkw), **bkw)
    torch.cuda.synchronize()
    hr = F.linear(a.float(), wo.float(), bo.float()) + h0.float() + vec.float()[vidx.to(dev)]
    yr = F.layer_norm(hr, (C,), gam.float(), bet.float(), 1e-5).half().float()
    hh, gg = F.linear(yr, w1.float(), b1.float()).chunk(2, dim=-1)
    ref = F.linear((hh * F.gelu(gg)).half().float(), w2.float(), b2.float()) + hr
    if mode == 2:
        ref = 0.37 * blend.float() + (1 - 0.37) * ref
    r12, r1, r2 = rel(one, comp), rel(one, ref), rel(comp, ref)
    print(f"fused projection + LN + feed-forward M={M} mode={mode}: vs composition {r12:.2e}; vs fp32 {r1:.2e} (composition {r2:.2e})")
    assert r1 < TOL and r1 < 1.05 * r2 + 1e-5 and r12 < 6e-4


def test_fused_feed_forward_race_screen(ops, dev):
    """The chunk ring of pt_ffn_geglu_f16 orders LDS-DMA landings against fragment reads by one counted vmcnt per phase and raw
    barriers; an early read passes a single check whenever the copy happened to land first.  A full chip of workgroups (eight
    rounds), many launches: every one bitwise identical to the first and to the two-launch form."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(11)
    M, C, I = 258048 // 4, 320, 1280
    x = h16(M, C, g=g, dev=dev)
    p1 = pack_linear(h16(2 * I, C, g=g, scale=C ** -0.5, dev=dev), h16(2 * I, g=g, scale=0.3, dev=dev), dev, geglu=True)
    p2 = pack_linear(h16(C, I, g=g, scale=I ** -0.5, dev=dev), h16(C, g=g, scale=0.3, dev=dev), dev)
    res = h16(M, C, g=g, dev=dev)
    two = ops.igemm(ops.igemm(x, p1), p2, res=res)
    out = torch.empty_like(two)
    for _ in range(40):
        ops.ffn_geglu(x, p1, p2, res=res, out=out)
        assert torch.equal(out, two)
    # the same for the form that starts at the attention output: its prologue has a ring (three out-projection buffers, counted
    # vmcnt), a LayerNorm exchange through LDS and the first W1 tiles copied in under it - every launch bitwise the first one
    po = pack_linear(h16(C, C, g=g, scale=C ** -0.5, dev=dev), h16(C, g=g, scale=0.2, dev=dev), dev)
    gam, bet = (1.0 + 0.2 * torch.randn(C, generator=g)).half().to(dev), (0.1 * torch.randn(C, generator=g)).half().to(dev)
    vec, blend = h16(2, C, g=g, dev=dev), h16(M, C, g=g, dev=dev)
    kw = dict(blend=blend, alpha=0.4, pre=dict(w=po, res=res, vec=vec, vec_mode=2, vFS=M // 2, vS=M // 28, vB=2, ln=(gam, bet, 1e-5)))
    first = ops.ffn_geglu(x, p1, p2, **kw).clone()
    for _ in range(40):
        ops.ffn_geglu(x, p1, p2, out=out, **kw)
        assert torch.equal(out, first)


# ------------------------------------------------------------------------------------------------- LayerNorm + linear in one launch (round 6)
@pytest.mark.parametrize("M,N,cs", [(128, 960, 320), (1000, 960, 320), (77, 960, 0), (4032, 320, 0), (258048 // 8 + 5, 960, 320), (130, 1024, 64),
                                    (256, 8, 0), (129, 72, 64)])
def test_ln_linear_equals_the_two_launches(ops, dev, M, N, cs):
    """pt_ln_linear_f16 (norm1 + attn1.to_q | to_k | to_v at C = 320, modified_svd.py:79-81) against pt_layernorm_f16 + pt_igemm_f16 and
    against fp32 torch, ragged and aligned M, N a multiple of 128 and not, with and without the pre-scaled Q columns.  The two forms
    round at the same points; the row statistics are summed in another order, so isolated values of LN(x) move by an fp16 ulp."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + N)
    K = 320
    x = (h16(M, K, g=g, dev=dev).float() * (1.0 + torch.rand(M, 1, generator=g).to(dev)) + 0.3 * torch.randn(M, 1, generator=g).to(dev)).half()
    w = h16(N, K, g=g, scale=K ** -0.5, dev=dev)
    gam, bet = (1.0 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev)
    pw = pack_linear(w, None, dev)
    assert ops.ln_linear_fusable(x, pw)
    kw = dict(cs_cols=cs, cs_scale=0.18) if cs else {}
    one = ops.ln_linear(x, gam, bet, pw, **kw)
    two = ops.igemm(ops.layernorm(x, gam, bet), pw, **kw)
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.float(), (K,), gam.float(), bet.float(), 1e-5).half().float(), w.float())
    if cs:
        ref[:, :cs] *= 0.18
    r1, r2, r12 = rel(one, ref), rel(two, ref), rel(one, two.float())
    assert one.shape == (M, N) and torch.isfinite(one).all()
    assert r1 < 4e-4 and r1 < 1.05 * r2 + 1e-5, (r1, r2)
    assert r12 < 2e-4, r12                                   # (measured ~3e-5: a handful of LN values one ulp apart)
    assert float((one != two).float().mean()) < 0.02
    # a strided output (a column block of a wider tensor) and an input with a row pitch
    wide_out = torch.zeros(M, N + 64, dtype=torch.float16, device=dev)
    xin = torch.zeros(M, K + 32, dtype=torch.float16, device=dev); xin[:, :K] = x
    ops.ln_linear(xin[:, :K], gam, bet, pw, out=wide_out[:, 64:64 + N], **kw)
    assert torch.equal(wide_out[:, 64:64 + N], one) and float(wide_out[:, :64].abs().max()) == 0.0
    o2 = ops.ln_linear(xin[:, :K], gam, bet, pw, **kw)
    assert torch.equal(o2, one)


def test_ln_linear_refuses_what_it_does_not_serve(ops, dev):
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(5)
    x = h16(64, 320, g=g, dev=dev)
    gam, bet = torch.ones(320).half().to(dev), torch.zeros(320).half().to(dev)
    with_bias = pack_linear(h16(320, 320, g=g, dev=dev), h16(320, g=g, dev=dev), dev)
    wide_k = pack_linear(h16(320, 640, g=g, dev=dev), None, dev)
    assert not ops.ln_linear_fusable(x, with_bias) and not ops.ln_linear_fusable(h16(64, 640, g=g, dev=dev), wide_k)
    with pytest.raises(RuntimeError):
        ops.ln_linear(x, gam, bet, with_bias)
    ok = pack_linear(h16(960, 320, g=g, dev=dev), None, dev)
    with pytest.raises(RuntimeError):                        # the column scale goes by wave halves of 64 columns
        ops.ln_linear(x, gam, bet, ok, cs_cols=8, cs_scale=2.0)


def test_ln_linear_race_screen(ops, dev):
    """The weight ring of pt_ln_linear_f16 orders LDS-DMA landings against fragment reads by one counted vmcnt per phase - with the output
    stores counted in - and raw barriers.  A full chip of workgroups (two rounds), many launches, every one bitwise the first."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(12)
    M, K, N = 258048 // 4, 320, 960
    x = h16(M, K, g=g, dev=dev)
    pw = pack_linear(h16(N, K, g=g, scale=K ** -0.5, dev=dev), None, dev)
    gam, bet = (1.0 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev)
    first = ops.ln_linear(x, gam, bet, pw, cs_cols=320, cs_scale=0.18).clone()
    two = ops.igemm(ops.layernorm(x, gam, bet), pw, cs_cols=320, cs_scale=0.18)
    assert rel(first, two.float()) < 2e-4
    out = torch.empty_like(first)
    for _ in range(40):
        ops.ln_linear(x, gam, bet, pw, cs_cols=320, cs_scale=0.18, out=out)
        assert torch.equal(out, first)


# ------------------------------------------------------------------------------------------------- convolutions
@pytest.mark.parametrize("N,H,W,Ci,Co,stride", [(2, 9, 16, 64, 128, 1), (3, 8, 8, 128, 64, 2), (2, 7, 5, 320, 320, 1),
                                                (1, 18, 32, 64, 64, 2), (2, 1, 1, 256, 256, 1)])
def test_conv3x3(ops, dev, N, H, W, Ci, Co, stride):
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(N * H + Ci)
    x = h16(N, H, W, Ci, g=g, dev=dev)
    w, b = h16(Co, Ci, 3, 3, g=g, scale=(9 * Ci) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
    y = ops.igemm(x, pack_conv2d(w, b, dev, stride=stride), geom=(N, H, W))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), stride=stride, padding=1).permute(0, 2, 3, 1)
    assert rel(y.view(ref.shape), ref) < TOL


def test_conv_small_channels_generic_path_and_silu(ops, dev):
    """Condition-encoder shapes: 3(->8 padded)->16 channels, SiLU epilogue, then 16->32 stride 2."""
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 24, 40
    x = (torch.rand(N, 3, H, W, generator=g) * 2 - 1).half().to(dev)
    w1, b1 = h16(16, 3, 3, 3, g=g, scale=27 ** -0.5, dev=dev), h16(16, g=g, dev=dev)
    w2, b2 = h16(32, 16, 3, 3, g=g, scale=144 ** -0.5, dev=dev), h16(32, g=g, dev=dev)
    p1, p2 = pack_conv2d(w1, b1, dev), pack_conv2d(w2, b2, dev, stride=2)
    p1.silu = p2.silu = True
    xcl = ops.to_channels_last(x, cpad=8)
    assert xcl.shape == (N, H, W, 8) and torch.all(xcl[..., 3:] == 0)
    y1 = ops.igemm(xcl, p1, geom=(N, H, W)).view(N, H, W, 16)
    y2 = ops.igemm(y1, p2, geom=(N, H, W)).view(N, H // 2, W // 2, 32)
    r1 = F.silu(F.conv2d(x.float(), w1.float(), b1.float(), padding=1))
    assert rel(y1.permute(0, 3, 1, 2), r1) < TOL
    r2 = F.silu(F.conv2d(y1.float().permute(0, 3, 1, 2), w2.float(), b2.float(), stride=2, padding=1))
    assert rel(y2.permute(0, 3, 1, 2), r2) < TOL


def test_conv_out_4_channels(ops, dev):
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(6)
    N, H, W, Ci = 2, 8, 8, 64
    x = h16(N, H, W, Ci, g=g, dev=dev)
    w, b = h16(4, Ci, 3, 3, g=g, scale=(9 * Ci) ** -0.5, dev=dev), h16(4, g=g, dev=dev)
    y = ops.igemm(x, pack_conv2d(w, b, dev), geom=(N, H, W))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert y.shape == (N * H * W, 4)
    assert rel(y.view(ref.shape), ref) < TOL


def test_conv_two_sources_1x1_and_3x3(ops, dev):
    """cat([hidden, skip], dim=C) folded into the gather (up blocks): 1x1 shortcut and 3x3."""
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(7)
    N, H, W, C0, C1, Co = 2, 6, 10, 128, 64, 128
    a, s = h16(N, H, W, C0, g=g, dev=dev), h16(N, H, W, C1, g=g, dev=dev)
    cat = torch.cat([a, s], dim=-1).float().permute(0, 3, 1, 2)
    for k in (1, 3):
        w, b = h16(Co, C0 + C1, k, k, g=g, scale=(k * k * (C0 + C1)) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
        y = ops.igemm(a, pack_conv2d(w, b, dev, padding=k // 2), x1=s, geom=(N, H, W))
        ref = F.conv2d(cat, w.float(), b.float(), padding=k // 2).permute(0, 2, 3, 1)
        assert rel(y.view(ref.shape), ref) < TOL, k


def test_conv_upsample2x(ops, dev):
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(8)
    N, H, W, C = 2, 5, 7, 64
    x = h16(N, H, W, C, g=g, dev=dev)
    w, b = h16(C, C, 3, 3, g=g, scale=(9 * C) ** -0.5, dev=dev), h16(C, g=g, dev=dev)
    y = ops.igemm(x, pack_conv2d(w, b, dev), geom=(N, H, W), upsample2x=True)
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(up, w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    assert rel(y.view(ref.shape), ref) < TOL


def test_conv_temporal_3x1x1(ops, dev):
    """Conv3d (3,1,1) over [B, C, F, H, W] == a (3 x 1) conv over the image (F, H*W) of the channels-last buffer."""
    from posetraj_amd.packing import pack_conv_t3
    g = torch.Generator().manual_seed(10)
    B, Fr, H, W, C = 2, 14, 3, 5, 128
    x = h16(B * Fr, H, W, C, g=g, dev=dev)
    w, b = h16(C, C, 3, 1, 1, g=g, scale=(3 * C) ** -0.5, dev=dev), h16(C, g=g, dev=dev)
    y = ops.igemm(x.view(B, Fr, H * W, C), pack_conv_t3(w, b, dev), geom=(B, Fr, H * W))
    x5 = x.float().view(B, Fr, H, W, C).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x5, w.float(), b.float(), padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * Fr * H * W, C)
    assert rel(y, ref) < TOL


# ------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("C0,C1,rows,ns,silu", [(320, 0, 72, 4, True), (64, 0, 64, 28, False), (1280, 640, 30, 3, True),
                                                (128, 64, 14 * 20, 2, True), (2560, 0, 9, 2, True)])
def test_groupnorm(ops, dev, C0, C1, rows, ns, silu):
    g = torch.Generator().manual_seed(C0 + rows)
    x0 = (torch.randn(ns * rows, C0, generator=g) * 1.5 + 0.7).half().to(dev)
    x1 = (torch.randn(ns * rows, C1, generator=g) * 0.5 - 1.0).half().to(dev) if C1 else None
    Ct = C0 + C1
    gamma, beta = h16(Ct, g=g, dev=dev), h16(Ct, g=g, dev=dev)
    y = ops.groupnorm(x0, gamma, beta, rows_per_sample=rows, n_samples=ns, eps=1e-5, silu=silu, x1=x1)
    xc = torch.cat([x0, x1], -1) if C1 else x0
    xr = xc.float().view(ns, rows, Ct).permute(0, 2, 1)                     # [ns, C, rows]
    ref = F.group_norm(xr, 32, gamma.float(), beta.float(), eps=1e-5)
    if silu:
        ref = F.silu(ref)
    assert rel(y.view(ns, rows, Ct), ref.permute(0, 2, 1)) < TOL


@pytest.mark.parametrize("M,C", [(1000, 320), (77, 640), (4032, 1280), (5, 64)])
def test_layernorm(ops, dev, M, C):
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.3).half().to(dev)
    gamma, beta = h16(C, g=g, dev=dev), h16(C, g=g, dev=dev)
    y = ops.layernorm(x, gamma, beta, 1e-5)
    assert rel(y, F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)) < TOL


def test_layernorm_with_frame_embedding(ops, dev):
    g = torch.Generator().manual_seed(4)
    N, S, C = 6, 21, 320
    x, e = h16(N * S, C, g=g, dev=dev), h16(N, C, g=g, dev=dev)
    gamma, beta = h16(C, g=g, dev=dev), h16(C, g=g, dev=dev)
    y = ops.layernorm(x, gamma, beta, 1e-5, vec=e, vG=S)
    xe = (x.view(N, S, C) + e[:, None, :]).float().view(N * S, C)           # fp16 add, like the reference
    assert rel(y, F.layer_norm(xe, (C,), gamma.float(), beta.float(), 1e-5)) < TOL


# ------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("Nimg,S,heads", [(2, 64, 1), (3, 144, 2), (2, 576, 5), (1, 2304, 2), (2, 45, 4), (1, 1, 4)])
def test_attn_spatial(ops, dev, Nimg, S, heads):
    g = torch.Generator().manual_seed(S + heads)
    C = heads * 64
    qkv = h16(Nimg * S, 3 * C, g=g, dev=dev)
    o = ops.attn_spatial(qkv, Nimg, S, heads, 64)
    q, k, v = [t.float().view(Nimg, S, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Nimg * S, C)
    assert rel(o, ref) < TOL_ATTN


def test_attn_spatial_forces_online_rescale(ops, dev):
    """One key far above the rest late in the sequence: the running max must jump and rescale O (guide rule 26)."""
    g = torch.Generator().manual_seed(12)
    S, heads, C = 320, 1, 64
    qkv = h16(S, 3 * C, g=g, scale=0.5, dev=dev)
    qkv[200, C:2 * C] = qkv[7, 0:C] * 6.0                   # key 200 aligned with query 7
    o = ops.attn_spatial(qkv, 1, S, heads, 64)
    q, k, v = [t.float().view(1, S, 1, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(S, C)
    assert rel(o, ref) < TOL_ATTN


@pytest.mark.parametrize("B,Fr,S,heads", [(2, 14, 64, 1), (1, 14, 45, 5), (2, 4, 9, 2), (2, 16, 7, 3), (3, 1, 5, 2),
                                          (2, 25, 36, 2), (1, 32, 9, 1), (2, 17, 5, 3)])   # F > 16: two 16-frame blocks (SVD-XT: 25)
@pytest.mark.parametrize("hd", [64, 128])
def test_attn_temporal(ops, dev, B, Fr, S, heads, hd):
    """head_dim 64 (SVD) and 128 (level 2 of the reference's in-tree default num_attention_heads = (5,10,10,20))."""
    g = torch.Generator().manual_seed(Fr + S)
    C = heads * hd
    qkv = h16(B * Fr * S, 3 * C, g=g, dev=dev)
    o = ops.attn_temporal(qkv, B, Fr, S, heads, hd)
    # reference: the permute/reshape dance of modified_svd.py:64-66,110-112
    def seq(t):
        return t.float().view(B, Fr, S, heads, hd).permute(0, 2, 3, 1, 4).reshape(B * S, heads, Fr, hd)
    q, k, v = [seq(t) for t in qkv.chunk(3, dim=-1)]
    r = F.scaled_dot_product_attention(q, k, v)                                          # [B*S, heads, F, hd]
    ref = r.view(B, S, heads, Fr, hd).permute(0, 3, 1, 2, 4).reshape(B * Fr * S, C)
    assert rel(o, ref) < TOL_ATTN


# ------------------------------------------------------------------------------------------------- element-wise
def test_axpy_silu_timestep_embedding(ops, dev):
    g = torch.Generator().manual_seed(13)
    a, r = h16(3, 37, 11, g=g, dev=dev), h16(3, 37, 11, g=g, dev=dev)
    assert rel(ops.axpy(a, r, 4.0), a.float() + 4.0 * r.float()) < 1e-3
    assert rel(ops.silu(a), F.silu(a.float())) < 1e-3
    t = torch.tensor([1.6378, -1.5537, 0.0, 6.0, 128.0, 0.02], device=dev)
    e = ops.timestep_embedding(t, 320)
    half = 160
    f = torch.exp(-math.log(10000.0) * torch.arange(half, device=dev) / half)
    ref = torch.cat([torch.cos(t[:, None] * f), torch.sin(t[:, None] * f)], -1)
    assert (e.float() - ref).abs().max() < 2e-3


def test_layout_round_trip(ops, dev):
    g = torch.Generator().manual_seed(14)
    x = h16(3, 20, 9, 13, g=g, dev=dev)                       # NCHW
    cl = ops.to_channels_last(x, cpad=24)
    assert torch.equal(cl[..., :20].permute(0, 3, 1, 2), x) and torch.all(cl[..., 20:] == 0)
    back = ops.to_nchw(cl, Cc=20)
    assert torch.equal(back, x)
    # zero-copy detection of a channels-last view
    v = cl[..., :24].permute(0, 3, 1, 2)
    assert ops.to_channels_last(v).data_ptr() == cl.data_ptr()
    assert rel(ops.to_channels_last(x.float()), x.permute(0, 2, 3, 1)) == 0.0


def test_scale_concat_and_cfg_euler(ops, dev):
    g = torch.Generator().manual_seed(15)
    Bc, Fr, h, w = 2, 5, 6, 7
    lat = (torch.randn(Bc, Fr, 4, h, w, generator=g) * 50).to(dev)
    il = h16(2 * Bc, 4, h, w, g=g, dev=dev)
    sigma, sigma_next = 12.5, 7.25
    xin = ops.scale_concat_input(lat, il, sigma)
    ref_lat = (torch.cat([lat] * 2) / (sigma ** 2 + 1) ** 0.5)
    ref = torch.cat([ref_lat, il.float()[:, None].repeat(1, Fr, 1, 1, 1)], dim=2)       # [2Bc, F, 8, h, w]
    assert rel(xin.permute(0, 1, 4, 2, 3), ref) < 1e-3
    pred = h16(2 * Bc, Fr, h, w, 4, g=g, dev=dev)
    guid = torch.linspace(1.0, 3.0, Fr).repeat(Bc, 1).to(dev)
    x = lat.clone()
    ops.cfg_euler_step(pred, guid, sigma, sigma_next, 0, x)
    p = pred.float().permute(0, 1, 4, 2, 3)
    un, co = p.chunk(2)
    mo = (un + guid[:, :, None, None, None] * (co - un)).half().float()
    x0 = mo * (-sigma / (sigma ** 2 + 1) ** 0.5) + lat / (sigma ** 2 + 1)
    refx = lat + (lat - x0) / sigma * (sigma_next - sigma)
    assert rel(x, refx) < 1e-5


def test_concat_camera(ops, dev):
    g = torch.Generator().manual_seed(16)
    feat, cam = h16(3, 4, 5, 64, g=g, dev=dev), h16(3, 12, g=g, dev=dev)
    y = ops.concat_camera(feat, cam, 80)
    assert torch.equal(y[..., :64], feat)
    assert torch.equal(y[..., 64:76], cam[:, None, None, :].expand(3, 4, 5, 12))
    assert torch.all(y[..., 76:] == 0)


def test_cpu_tensors_are_refused(ops):
    from posetraj_amd.packing import pack_linear
    with pytest.raises(RuntimeError):
        ops.silu(torch.zeros(8, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 64, dtype=torch.float16), torch.ones(64).half(), torch.zeros(64).half())


# ------------------------------------------------------------------------------------------------- split-K
@pytest.mark.parametrize("Nimg,H,W,Ci,Co", [(28, 9, 16, 1280, 1280), (28, 5, 9, 1280, 1280), (6, 9, 16, 2560, 1280)])
def test_splitk_conv_small_m(ops, dev, Nimg, H, W, Ci, Co):
    """Level-3 shapes (M = 4032 / 1260 rows): the 256 x 320 kernel deals the K tiles to several workgroups per output tile
    and a second launch adds the fp32 slabs in order and runs the epilogue.  Same result as the single launch up to the
    summation order, bit-identical between two runs."""
    from posetraj_amd.packing import pack_conv2d
    g = torch.Generator().manual_seed(Nimg + H + Ci)
    x = h16(Nimg, H, W, Ci, g=g, dev=dev)
    w, b = h16(Co, Ci, 3, 3, g=g, scale=(9 * Ci) ** -0.5, dev=dev), h16(Co, g=g, dev=dev)
    res, vec = h16(Nimg * H * W, Co, g=g, dev=dev), h16(2, Co, g=g, dev=dev)
    pw = pack_conv2d(w, b, dev)
    kw = dict(geom=(Nimg, H, W), res=res, vec=vec, vec_mode=1, vG=Nimg * H * W // 2)
    y = ops.igemm(x, pw, **kw)
    y1 = ops.igemm(x, pw, splitk=False, **kw)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    ref = ref + res.float() + vec.float().repeat_interleave(Nimg * H * W // 2, dim=0)
    assert rel(y, ref) < 6e-4 and rel(y1, ref) < 6e-4
    assert rel(y, y1.float()) < 3e-4
    assert torch.equal(y, ops.igemm(x, pw, **kw))


def test_splitk_linear_full_epilogue(ops, dev):
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(77)
    M, N, K = 2520, 1280, 5120          # (the training step's level-2 row count: 200 tiles of 128 x 128, so plan_splits splits; since
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)   # round 6 4032 rows stay un-split up to K = 5120)
    res, blend = h16(M, N, g=g, dev=dev), h16(M, N, g=g, dev=dev)
    pw = pack_linear(w, b, dev)
    import ctypes as C
    y = ops.igemm(x, pw, res=res, blend=blend, alpha=0.3, out_scale=0.5, cs_cols=640, cs_scale=2.0)
    ref = 0.5 * (0.3 * blend.float() + 0.7 * (F.linear(x.float(), w.float(), b.float()) + res.float()))
    ref[:, :640] *= 2.0
    assert rel(y, ref) < 6e-4
    # accumulate-into-residual form (the fused ControlNet zero-conv)
    acc = res.clone()
    ops.igemm(x, pw, res=acc, res_post=True, out_scale=3.0, out=acc)
    assert rel(acc, res.float() + 3.0 * F.linear(x.float(), w.float(), b.float())) < 6e-4


@pytest.mark.parametrize("M,N,K", [(1000, 1280, 11520), (768, 1280, 5120), (512, 1280, 8640)])
def test_splitk_uneven_slices(ops, dev, M, N, K):
    """Split counts that do not divide the K tiles (nk = 180 with <= 16 tiles planned 16 splits of 12 tiles - the last one
    empty; nk = 80: 13 x 7 > 80; nk = 135): plan_splits now drops the empty slices; the result must equal the unsplit
    launch up to the summation order either way."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(M + K)
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    res = h16(M, N, g=g, dev=dev)
    pw = pack_linear(w, b, dev)
    y = ops.igemm(x, pw, res=res)
    y1 = ops.igemm(x, pw, res=res, splitk=False)
    ref = F.linear(x.float(), w.float(), b.float()) + res.float()
    assert rel(y, ref) < 6e-4 and rel(y1, ref) < 6e-4 and rel(y, y1.float()) < 3e-4
    assert torch.equal(y, ops.igemm(x, pw, res=res))


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4])
def test_wide_stream_pair_output_and_pair_residual(ops, dev, cfg):
    """Residual-stream tensors as fp16 pairs: ``out`` is exactly the plain fp16 result, ``out + out.lo`` carries the value
    to ~2^-22, and a residual that has a low half is added as the pair (every tile configuration; conv and linear;
    split-K reducer)."""
    from posetraj_amd import hip
    from posetraj_amd.packing import pack_conv2d, pack_linear
    g = torch.Generator().manual_seed(5 + cfg)
    M, N, K = 4096 + 64, 640, 320
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    r_hi = h16(M, N, g=g, dev=dev)
    r_lo = (h16(M, N, g=g, dev=dev) * 2.0 ** -12).half()
    res = r_hi.clone(); res.lo = r_lo
    hip.check(hip.lib().pt_igemm_force_config(cfg))
    try:
        pw = pack_linear(w, b, dev)
        ref = F.linear(x.double(), w.double(), b.double()) + r_hi.double() + r_lo.double()
        y = ops.igemm(x, pw, res=res, wide=True)
        assert y.lo is not None and torch.equal(y, ops.igemm(x, pw, res=res))          # high half == the plain result
        assert rel(y.double() + y.lo.double(), ref) < 4e-6                             # fp32 accumulation noise only
        assert 1e-4 < rel(y, ref) < 3e-4                                               # one fp16 rounding without the pair
        y2 = ops.igemm(x, pw, res=res, res_post=True, out_scale=0.25, wide=True)
        ref2 = 0.25 * F.linear(x.double(), w.double(), b.double()) + r_hi.double() + r_lo.double()
        assert rel(y2.double() + y2.lo.double(), ref2) < 4e-6
        y0 = ops.igemm(x, pw, wide=True)                                                # no side input (shortcut conv)
        assert rel(y0.double() + y0.lo.double(), F.linear(x.double(), w.double(), b.double())) < 4e-6
    finally:
        hip.check(hip.lib().pt_igemm_force_config(-1))
    if cfg == -1:                                                                       # split-K reducer (level-3 conv)
        xc = h16(28, 9, 16, 1280, g=g, dev=dev)
        wc, bc = h16(1280, 1280, 3, 3, g=g, scale=(9 * 1280) ** -0.5, dev=dev), h16(1280, g=g, dev=dev)
        rc = h16(28 * 9 * 16, 1280, g=g, dev=dev); rc.lo = (h16(28 * 9 * 16, 1280, g=g, dev=dev) * 2.0 ** -12).half()
        yc = ops.igemm(xc, pack_conv2d(wc, bc, dev), geom=(28, 9, 16), res=rc, wide=True)
        refc = F.conv2d(xc.double().permute(0, 3, 1, 2), wc.double(), bc.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, 1280)
        refc = refc + rc.double() + rc.lo.double()
        assert rel(yc.double() + yc.lo.double(), refc) < 6e-6


# ------------------------------------------------------------------------------------------------- implementation error
def _impl(y, ref64):
    """rel-L2 of the kernel output against the fp16 rounding of the exact result: what the kernel adds on top of the ONE
    rounding any fp16-storing kernel must make (tools/op_ladder.py prints the same split for every op)."""
    r16 = ref64.half().double()
    return rel(y, r16), rel(r16, ref64)


def test_kernels_are_exact_up_to_their_output_rounding(ops, dev):
    from posetraj_amd.packing import pack_conv2d, pack_linear
    g = torch.Generator().manual_seed(123)
    D = lambda t: t.double().cpu()
    M, N, K = 2048, 320, 1280
    x, w, b = h16(M, K, g=g, dev=dev), h16(N, K, g=g, scale=K ** -0.5, dev=dev), h16(N, g=g, dev=dev)
    res = h16(M, N, g=g, dev=dev)
    impl, out = _impl(ops.igemm(x, pack_linear(w, b, dev), res=res), F.linear(D(x), D(w), D(b)) + D(res))
    assert impl < 5e-5 and 1.5e-4 < out < 2.5e-4, (impl, out)
    wg, bg = h16(2560, 320, g=g, scale=320 ** -0.5, dev=dev), h16(2560, g=g, dev=dev)
    xg = h16(M, 320, g=g, dev=dev)
    hh, gg = F.linear(D(xg), D(wg), D(bg)).chunk(2, dim=-1)
    impl, _ = _impl(ops.igemm(xg, pack_linear(wg, bg, dev, geglu=True)), hh * F.gelu(gg))
    assert impl < 5e-5, impl
    xc = h16(2, 24, 32, 320, g=g, dev=dev)
    wc, bc = h16(320, 320, 3, 3, g=g, scale=2880 ** -0.5, dev=dev), h16(320, g=g, dev=dev)
    impl, _ = _impl(ops.igemm(xc, pack_conv2d(wc, bc, dev), geom=(2, 24, 32)).view(2, 24, 32, 320),
                    F.conv2d(D(xc).permute(0, 3, 1, 2), D(wc), D(bc), padding=1).permute(0, 2, 3, 1))
    assert impl < 5e-5, impl
    xn = (torch.randn(4 * 768, 320, generator=g) * 1.5 + 0.4).half().to(dev)
    ga, be = (torch.randn(320, generator=g) * 0.1 + 1).half().to(dev), h16(320, g=g, scale=0.1, dev=dev)
    y = ops.groupnorm(xn, ga, be, rows_per_sample=768, n_samples=4, eps=1e-5, silu=True)
    ref = F.silu(F.group_norm(D(xn).view(4, 768, 320).permute(0, 2, 1), 32, D(ga), D(be), eps=1e-5)).permute(0, 2, 1).reshape(-1, 320)
    impl, _ = _impl(y, ref)
    assert impl < 3e-5, impl
    impl, _ = _impl(ops.layernorm(xn, ga, be, 1e-5), F.layer_norm(D(xn), (320,), D(ga), D(be), 1e-5))
    assert impl < 3e-5, impl
    # attention carries its P matrix in fp16 (one extra rounding per probability, averaged over the keys)
    S, heads = 2304, 5
    qkv = h16(S, 3 * heads * 64, g=g, dev=dev)
    qkv[:, :2 * heads * 64] *= 0.35
    q, k, v = [D(t).view(1, S, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(S, heads * 64)
    impl, _ = _impl(ops.attn_spatial(qkv, 1, S, heads, 64), ref)
    assert impl < 3e-4, impl
    qs = qkv.clone()
    qs[:, :heads * 64] = (qs[:, :heads * 64].float() * ops.attn_q_prescale(64)).half()
    impl2, _ = _impl(ops.attn_spatial(qs, 1, S, heads, 64, q_prescaled=True), ref)
    assert impl2 < 4e-4, impl2


def test_attn_spatial_threshold_crossings(ops, dev):
    """Defer-max softmax: the reference max is re-based only when a tile's maximum exceeds it by 2^8.  Keys whose scores
    climb steadily force a crossing every few tiles; keys far BELOW the reference must not disturb it (guide rule 26)."""
    g = torch.Generator().manual_seed(21)
    S, C = 1024, 64
    qkv = h16(S, 3 * C, g=g, scale=0.3, dev=dev)
    ramp = torch.linspace(0.0, 40.0, S, device=dev).half()                     # score of key j grows with j for every query:
    qkv[:, 0] = 4.0                                                             # q[:, 0] = 4, k[j, 0] = ramp_j -> + 4 ramp_j / 8
    qkv[:, C] = ramp                                                            # = 29 log2-units over 16 tiles: a crossing every ~4
    qkv[::3, C] = -40.0                                                         # every third key 29 log2-units below
    for pre in (False, True):
        t = qkv.clone()
        if pre:
            t[:, :C] = (t[:, :C].float() * ops.attn_q_prescale(64)).half()
        o = ops.attn_spatial(t, 1, S, 1, 64, q_prescaled=pre)
        q, k, v = [x.float().view(1, S, 1, 64).transpose(1, 2) for x in qkv.chunk(3, dim=-1)]
        ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(S, C)
        assert rel(o, ref) < (2e-3 if pre else TOL), (pre, rel(o, ref))


def test_stale_wide_pair_is_refused(ops, dev):
    """a residual-stream pair whose high half is rewritten in place (igemm(out=...)) must not be added back with
    its old low half - the next use as `res` raises unless the writer dropped the low half (ops.drop_lo)."""
    from posetraj_amd.packing import pack_linear
    g = torch.Generator().manual_seed(3)
    x, w = h16(256, 64, g=g, dev=dev), h16(64, 64, g=g, scale=0.125, dev=dev)
    pw = pack_linear(w, None, dev)
    y = ops.igemm(x, pw, wide=True)
    assert hasattr(y, "lo")
    z = ops.igemm(x, pw, res=y)                                              # fresh pair: fine
    ops.igemm(x, pw, out=y)                                                  # in-place rewrite of the high half
    with pytest.raises(RuntimeError, match="stale fp16 pair"):
        ops.igemm(x, pw, res=y)
    ops.drop_lo(y)
    z2 = ops.igemm(x, pw, res=y)                                             # plain fp16 residual now
    assert torch.isfinite(z2).all() and z.shape == z2.shape
