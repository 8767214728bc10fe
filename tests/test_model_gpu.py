"""Model-level parity on the MI355X: the drop-in classes of posetraj_amd (HIP path through the C ABI) against the CPU
oracle on identical fp16-representable weights and inputs; and against the golden fixtures produced by the reference."""
import numpy as np
import pytest
import torch

from tests import parity as P

pytestmark = pytest.mark.gpu
# north-star target for the whole path is 1e-3 rel-L2; individual residual taps and the tiny random net are held to:
TOL_NET = 2.2e-3        # r01: 5e-3.  Measured 1.4e-3 .. 1.8e-3 for the deepest quantity (a CFG loop iteration); the ladder
# that splits this into dtype and implementation is asserted in tests/test_parity_ladder_gpu.py


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return "cuda:0"


@pytest.fixture(scope="module")
def nets(dev):
    cn_o, unet_o = P.build_oracle_nets(seed=0)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    return cn_o, unet_o, cn_h, unet_h


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


def test_controlnet_forward(nets, dev):
    cn_o, _, cn_h, _ = nets
    i = P.tiny_inputs(seed=1)
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False,
                             conditioning_scale=0.7)
    j = _to(i, dev)
    down_h, mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(),
                         return_dict=False, conditioning_scale=0.7)
    assert len(down_h) == 12
    for a, b in zip(down_h, down_o):
        assert tuple(a.shape) == tuple(b.shape)
        assert P.rel_l2(a, b) < TOL_NET
    assert P.rel_l2(mid_h, mid_o) < TOL_NET
    # dict-style return
    out = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), conditioning_scale=0.7)
    assert torch.equal(out.mid_block_res_sample, mid_h)
    assert len(out.down_block_res_samples) == 12 and torch.equal(out[0][3], down_h[3])


def test_controlnet_without_condition_and_float_timestep(nets, dev):
    cn_o, _, cn_h, _ = nets
    i = P.tiny_inputs(seed=2)
    with torch.no_grad():
        _, mid_o = cn_o(i["sample"], 1.137, i["ehs"], i["ids"], controlnet_cond=None, return_dict=False)
    j = _to(i, dev)
    _, mid_h = cn_h(j["sample"].half(), 1.137, j["ehs"].half(), j["ids"], controlnet_cond=None, return_dict=False)
    assert P.rel_l2(mid_h, mid_o) < TOL_NET


def test_unet_forward_with_residual_multiplicity(nets, dev):
    cn_o, unet_o, cn_h, unet_h = nets
    i = P.tiny_inputs(seed=3)
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)
        y_o = unet_o(i["sample"], i["t"], i["ehs"], down_o, mid_o, return_dict=False, added_time_ids=i["ids"])[0]
    j = _to(i, dev)
    # (a) residuals produced by the HIP ControlNet (channels-last views, zero-copy hand-over)
    down_h, mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(),
                         return_dict=False)
    y_h = unet_h(j["sample"].half(), j["t"], j["ehs"].half(), down_h, mid_h, return_dict=False, added_time_ids=j["ids"])[0]
    assert tuple(y_h.shape) == tuple(y_o.shape)
    assert P.rel_l2(y_h, y_o) < TOL_NET
    # (b) residuals handed over as ordinary contiguous NCHW tensors from the oracle
    y_h2 = unet_h(j["sample"].half(), j["t"], j["ehs"].half(), [d.half().to(dev) for d in down_o], mid_o.half().to(dev),
                  return_dict=False, added_time_ids=j["ids"])[0]
    assert P.rel_l2(y_h2, y_o) < TOL_NET
    # (c) dropping the multiplicity would be visible: scale residuals by 1 instead of (4,4,4,4,3,...) -> different output
    with torch.no_grad():
        y_wrong = unet_o(i["sample"], i["t"], i["ehs"], [d / m for d, m in zip(down_o, (4, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1))],
                         mid_o, return_dict=False, added_time_ids=i["ids"])[0]
    assert P.rel_l2(y_wrong, y_o) > 10 * P.rel_l2(y_h, y_o)


def test_wide_residual_stream_switch(nets, dev):
    """ops.WIDE_STREAM: the fp16-pair residual stream is what brings the U-Net forward under 1e-3 of the fp32
    oracle; with it off the same kernels store single fp16 tensors (round 1's 1.1e-3 .. 1.2e-3) and nothing else changes."""
    from posetraj_amd import ops
    cn_o, unet_o, cn_h, unet_h = nets
    i = P.tiny_inputs(seed=3, h=16, w=16)
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False)
        y_o = unet_o(i["sample"], i["t"], i["ehs"], down_o, mid_o, return_dict=False, added_time_ids=i["ids"])[0]
    j = _to(i, dev)
    res = [d.half().to(dev) for d in down_o], mid_o.half().to(dev)
    run = lambda: unet_h(j["sample"].half(), j["t"], j["ehs"].half(), res[0], res[1], return_dict=False, added_time_ids=j["ids"])[0]
    assert ops.WIDE_STREAM
    r_wide = P.rel_l2(run(), y_o)
    ops.WIDE_STREAM = False
    try:
        r_plain = P.rel_l2(run(), y_o)
    finally:
        ops.WIDE_STREAM = True
    assert r_wide < 1.0e-3, r_wide                 # measured 7.8e-4
    assert r_wide < 0.85 * r_plain < 1.5e-3, (r_wide, r_plain)      # measured 7.8e-4 vs 1.14e-3


def test_unet_requires_residuals(nets, dev):
    _, _, _, unet_h = nets
    j = _to(P.tiny_inputs(seed=4), dev)
    with pytest.raises(TypeError):
        unet_h(j["sample"].half(), j["t"], j["ehs"].half(), added_time_ids=j["ids"])


def test_camera_controlnet(dev):
    cn_o, unet_o = P.build_oracle_nets(seed=10, camera=True)
    cn_h, _ = P.build_hip_nets(cn_o, unet_o, dev, camera=True)
    i = P.tiny_inputs(seed=5)
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], camera_cond=i["cam"],
                             return_dict=False)
    j = _to(i, dev)
    down_h, mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(),
                         camera_cond=j["cam"].half(), return_dict=False)
    assert P.rel_l2(down_h[0], down_o[0]) < TOL_NET
    assert P.rel_l2(mid_h, mid_o) < TOL_NET


@pytest.mark.parametrize("camera", [False, True])
def test_pipeline_two_steps(dev, camera):
    r, out, ref = P.run_tiny_pipeline_parity(steps=2, device=dev, camera=camera, return_all=True)
    assert out.shape == ref.shape == (1, 14, 4, 8, 8)
    assert r < TOL_NET, r


@pytest.mark.parametrize("frames", [25, 17])
def test_pipeline_more_than_16_frames(dev, frames):
    """SVD-XT's 25 frames (also the in-tree default num_frames = 25, models/controlnet_sdv.py:263) and an odd 17: the temporal
    attention runs two 16-frame blocks, the temporal convolutions / GroupNorms see F x H x W rows, Q3's interleave as before."""
    r, out, ref = P.run_tiny_pipeline_parity(steps=1, device=dev, frames=frames, latent_hw=(8, 8), return_all=True)
    assert out.shape == ref.shape == (1, frames, 4, 8, 8)
    assert r < TOL_NET, r


def test_pipeline_ragged_latent(dev):
    """Non-square latent whose token counts are not multiples of the attention tiles (S = 128, 32, 8, 2).  Sizes that
    go odd through the stride-2 convs are rejected by the reference U-Net itself (skip / upsample shape mismatch)."""
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(8, 16), device=dev)
    assert r < TOL_NET, r


def test_scheduler_device_steps_match_reference_goldens(golden, dev):
    from posetraj_amd import EulerDiscreteScheduler, SVD_SCHEDULER_CONFIG
    g = golden("sched")
    for n in (2, 25):
        k = f"svd_n{n}_"
        s = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
        s.set_timesteps(n, device=dev)
        x = torch.from_numpy(g[k + "x0_f32"]).to(dev)
        for i in range(2):
            t = s.timesteps[i]
            xin = s.scale_model_input(x, t)
            assert np.allclose(xin.cpu().numpy(), g[k + f"scaled{i}_f32"], rtol=2e-6, atol=0)
            mo = torch.from_numpy(g[k + f"model_out{i}_f32"]).to(dev)
            x = s.step(mo, t, x).prev_sample
            ref = g[k + f"prev{i}_f32"]
            # the update cancels a sigma-scale sample (|x| ~ 700 * N(0,1)) down to the result's scale, so agreement is
            # bounded by a few fp32 ulps of the INPUT magnitude, not of the result
            xin_max = float(np.abs(g[k + (f"prev{i - 1}_f32" if i else "x0_f32")]).max())
            assert float(np.abs(x.cpu().numpy() - ref).max()) <= 4 * np.finfo(np.float32).eps * xin_max
            x = torch.from_numpy(ref).to(dev)
        with pytest.raises(ValueError):
            s.step(mo, 3, x)


def test_save_pretrained_from_pretrained_round_trip(tmp_path, dev):
    """diffusers directory layout (config.json + diffusion_pytorch_model.safetensors): what the reference's callers
    load with ControlNetSDVModel.from_pretrained(ckpt, subfolder="controlnet") (scripts/run_inference...:335-337)."""
    from posetraj_amd.controlnet_sdv import ControlNetSDVModel
    from posetraj_amd.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel
    cn_o, unet_o = P.build_oracle_nets(seed=3)
    cn = ControlNetSDVModel(**P.TINY, conditioning_embedding_out_channels=P.TINY_CE).load_state_dict(
        cn_o.state_dict(), dev, keep_source=True)
    unet = UNetSpatioTemporalConditionControlNetModel(**P.TINY).load_state_dict(unet_o.state_dict(), dev, keep_source=True)
    cn.save_pretrained(str(tmp_path / "ckpt" / "controlnet"))
    unet.save_pretrained(str(tmp_path / "ckpt" / "unet"))
    cn2 = ControlNetSDVModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="controlnet", device=dev)
    unet2 = UNetSpatioTemporalConditionControlNetModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="unet", device=dev)
    assert dict(cn2.config) == dict(cn.config) and dict(unet2.config) == dict(unet.config)
    j = {k: v.to(dev) for k, v in P.tiny_inputs(seed=6).items()}
    a = cn(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), return_dict=False)
    b = cn2(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), return_dict=False)
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])) and torch.equal(a[1], b[1])
    ya = unet(j["sample"].half(), j["t"], j["ehs"].half(), a[0], a[1], return_dict=False, added_time_ids=j["ids"])[0]
    yb = unet2(j["sample"].half(), j["t"], j["ehs"].half(), b[0], b[1], return_dict=False, added_time_ids=j["ids"])[0]
    assert torch.equal(ya, yb)


def test_from_unet_copies_encoder_and_starts_as_noop(dev):
    """ControlNetSDVModel.from_unet (controlnet_sdv.py:653-709): encoder weights copied (not add_embedding), output
    convs zero -> the fresh ControlNet contributes exact zeros."""
    from posetraj_amd.controlnet_sdv import ControlNetSDVModel
    from posetraj_amd.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel
    _, unet_o = P.build_oracle_nets(seed=4)
    unet = UNetSpatioTemporalConditionControlNetModel(**P.TINY).load_state_dict(unet_o.state_dict(), dev, keep_source=True)
    cn = ControlNetSDVModel.from_unet(unet, conditioning_embedding_out_channels=P.TINY_CE)
    sd_u, sd_c = unet.state_dict(), cn.state_dict()
    copied = {k.split(".")[0] for k in sd_c if k in sd_u and torch.equal(sd_c[k], sd_u[k])}
    assert {"conv_in", "time_embedding", "down_blocks", "mid_block"} <= copied
    assert not any(k.startswith("add_embedding") and torch.equal(sd_c[k], sd_u[k]) for k in sd_c if k in sd_u)
    j = {k: v.to(dev) for k, v in P.tiny_inputs(seed=7).items()}
    down, mid = cn(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(), return_dict=False)
    assert all(float(d.abs().max()) == 0.0 for d in down) and float(mid.abs().max()) == 0.0


def test_hipgraph_replay_equals_eager_and_is_reused_across_clips(dev):
    """denoise(use_graph=True) captures ControlNet + U-Net once and replays it per iteration; results must equal the
    eager launches bit for bit, also for a second clip (new latents / embedding / control maps) on the same graph."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    from tools.variants.split_cfg import networks_split
    cn_o, unet_o = P.build_oracle_nets(seed=8)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h,
                                                  scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    outs = {}
    for clip in (0, 1):
        g = torch.Generator().manual_seed(100 + clip)
        lat = (torch.randn(1, 14, 4, 8, 8, generator=g) * 700).to(dev)
        mode = torch.randn(1, 4, 8, 8, generator=g).half()
        il = torch.cat([torch.zeros_like(mode), mode]).to(dev)
        e = torch.randn(1, 1, 64, generator=g).half()
        emb = torch.cat([torch.zeros_like(e), e]).to(dev)
        c1 = (torch.rand(1, 14, 3, 64, 64, generator=g) * 2 - 1).half()
        cond = torch.cat([c1, c1]).to(dev)
        for mode_name, ug, ov in (("eager", False, False), ("graph", True, False), ("eager2", False, True), ("graph2", True, True)):
            outs[(clip, mode_name)] = pipe.denoise(lat, il, emb, cond, num_inference_steps=3, use_graph=ug, overlap_streams=ov)
        assert torch.equal(outs[(clip, "eager")], outs[(clip, "graph")])
        # ControlNet and the U-Net's encoder half on two HIP streams: same kernels, bit-identical results
        assert torch.equal(outs[(clip, "eager")], outs[(clip, "eager2")])
        assert torch.equal(outs[(clip, "eager")], outs[(clip, "graph2")])
    assert pipe._graph_state is not None and "graph" in pipe._graph_state
    assert not torch.equal(outs[(0, "graph")], outs[(1, "graph")])


def test_condition_encoder_against_reference_golden(golden, dev):
    """The HIP condition encoder (8 igemm launches with fused SiLU, camera concat + per-pixel Linear) against
    tests/golden/cond_embed.npz - outputs of the REFERENCE classes (models/controlnet_sdv.py:61-116,
    controlnet_sdv_cam_infer.py:61-130) on seeded weights; the oracle is not involved."""
    from oracle import cond_embed as OC, init as OI
    from posetraj_amd.controlnet_sdv import ControlNetConditioningEmbeddingSVD
    g = golden("cond_embed")
    for camera, seed, cls in ((False, 21, OC.ControlNetConditioningEmbeddingSVD), (True, 22, OC.ControlNetConditioningEmbeddingSVD_CAM)):
        sd = {"e." + k: v for k, v in OI.seeded_init_(cls(64, 3, (8, 16, 32, 64)), seed=seed).state_dict().items()}
        sd = {k: v.half().float() for k, v in sd.items()}
        enc = ControlNetConditioningEmbeddingSVD(sd, "e.", dev, camera)
        ref_mod = cls(64, 3, (8, 16, 32, 64))
        ref_mod.load_state_dict({k[2:]: v for k, v in sd.items()})
        for b in (1, 2):
            x = torch.from_numpy(g[f"x_b{b}"]).half()
            rt = torch.from_numpy(g[f"rt_b{b}"]).half()
            cases = [("y", None)] if not camera else [("ycam", rt), ("ycam_none", None), ("ycam_zero", torch.zeros_like(rt))]
            for key, cam in cases:
                y = enc.run(x.to(dev), None if cam is None else cam.to(dev), None).permute(0, 3, 1, 2)
                want = torch.from_numpy(g[f"{key}_b{b}"])
                # the golden was made with fp32 weights / inputs; both sides here use their fp16 roundings, so the
                # bound is the fp16 input quantisation (measured 6e-4) - and the same module on the rounded values
                # must agree to the kernel tolerance
                with torch.no_grad():
                    same = ref_mod(x.float(), cam.float()) if (camera and cam is not None) else (ref_mod(x.float(), None) if camera else ref_mod(x.float()))
                assert P.rel_l2(y, same) < 1e-3, (camera, b, key, P.rel_l2(y, same))
                assert P.rel_l2(y, want) < 3e-3, (camera, b, key, P.rel_l2(y, want))


def test_pipeline_fused_residuals_equal_the_public_api_path(nets, dev):
    """The pipeline accumulates  multiplicity x scale x zero_conv(tap)  straight into the U-Net skips (res_post epilogue)
    and asks conv_out for fp32; the public forwards hand the residuals over as tensors and add them with pt_axpy_f16.
    Both are the same arithmetic up to the roundings the fused path removes."""
    cn_o, unet_o, cn_h, unet_h = nets
    j = _to(P.tiny_inputs(seed=9), dev)
    down, mid = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(),
                     return_dict=False, conditioning_scale=0.8)
    y_api = unet_h(j["sample"].half(), j["t"], j["ehs"].half(), down, mid, return_dict=False, added_time_ids=j["ids"])[0]
    enc = unet_h._encode(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"])
    taps, xm = cn_h._features(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], j["cond"].half(), None)
    mult = unet_h._multiplicity(enc, len(taps))
    assert mult == [4, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1]
    cn_h._accumulate_into(taps, xm, 0.8, enc["skips"], mult, enc["x"])
    y_fused = unet_h._decode(enc, None, None, return_dict=False, residuals_added=True, out_f32=True)[0]
    assert y_fused.dtype == torch.float32 and tuple(y_fused.shape) == tuple(y_api.shape)
    assert P.rel_l2(y_fused, y_api) < 1.5e-3
    i = P.tiny_inputs(seed=9)
    with torch.no_grad():
        d_o, m_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], return_dict=False, conditioning_scale=0.8)
        y_o = unet_o(i["sample"], i["t"], i["ehs"], d_o, m_o, return_dict=False, added_time_ids=i["ids"])[0]
    assert P.rel_l2(y_fused, y_o) <= P.rel_l2(y_api, y_o) * 1.05        # fewer roundings: not worse than the API path


def test_hipgraph_is_reused_for_a_second_clip(dev):
    """Two different clips back to back with identical flags: the graph object is captured once, replayed for both,
    and each result equals the eager launches (the reuse path must be exercised, not only the capture)."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    cn_o, unet_o = P.build_oracle_nets(seed=12)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h,
                                                  scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    graphs, res = [], []
    for clip in (0, 1, 2):
        g = torch.Generator().manual_seed(300 + clip)
        lat = (torch.randn(1, 14, 4, 8, 8, generator=g) * 700).to(dev)
        mode = torch.randn(1, 4, 8, 8, generator=g).half()
        il = torch.cat([torch.zeros_like(mode), mode]).to(dev)
        e = torch.randn(1, 1, 64, generator=g).half()
        emb = torch.cat([torch.zeros_like(e), e]).to(dev)
        c1 = (torch.rand(1, 14, 3, 64, 64, generator=g) * 2 - 1).half()
        cond = torch.cat([c1, c1]).to(dev)
        out_g = pipe.denoise(lat, il, emb, cond, num_inference_steps=2, use_graph=True, overlap_streams=True)
        graphs.append(pipe._graph_state["graph"])
        if clip == 1:
            # an eager ControlNet call with ANOTHER condition geometry between replays replaces the ControlNet's cached
            # condition embedding; the graph owns its own embedding buffer, so the next replay must still be right
            big = (torch.rand(2, 14, 3, 128, 128, generator=g) * 2 - 1).half().to(dev)
            lat2 = torch.randn(2, 14, 8, 16, 16, generator=g).half().to(dev)
            cn_h(lat2, torch.tensor(1.0), emb, torch.tensor([[6, 128, 0.02]] * 2).to(dev), controlnet_cond=big, return_dict=False)
        out_e = pipe.denoise(lat, il, emb, cond, num_inference_steps=2, use_graph=False, overlap_streams=False)
        assert torch.equal(out_g, out_e), clip
        res.append(out_g)
    assert graphs[0] is graphs[1] and graphs[1] is graphs[2]     # captured once, replayed for every clip
    assert not torch.equal(res[0], res[1])


@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_add_noise_matches_reference_golden(golden, dev, tag, dt):
    """EulerDiscreteScheduler.add_noise on the device, against the reference's outputs, bit for bit."""
    from posetraj_amd import EulerDiscreteScheduler, SVD_SCHEDULER_CONFIG
    g = golden("add_noise")
    s = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
    s.set_timesteps(25, device=dev)
    y = s.add_noise(torch.from_numpy(g[f"svd_{tag}_x"]).to(dev, dt), torch.from_numpy(g[f"svd_{tag}_noise"]).to(dev, dt),
                    torch.from_numpy(g[f"svd_{tag}_t"]).to(dev))
    assert y.dtype == dt
    assert np.array_equal(y.float().cpu().numpy(), g[f"svd_{tag}_y"])
    with pytest.raises(ValueError):
        s.add_noise(torch.zeros(1, 4, device=dev), torch.zeros(1, 4, device=dev), torch.tensor([123.456]))


# ------------------------------------------------------------------------------------------------- pre-loop stages
@pytest.mark.parametrize("name", ["down_L", "down_frac", "up", "chw", "clip224"])
def test_resize_with_antialiasing_against_the_reference_golden(dev, golden, name):
    """pt_resize_antialias_f32 (Gaussian blur with reflect padding + bicubic, align_corners) against outputs of the
    reference's _resize_with_antialiasing (tests/golden/resize.npz); fp32 both sides, tolerance = summation order."""
    from posetraj_amd import ops
    g = golden("resize")
    x = torch.from_numpy(g[name + "_x"]).to(dev)
    y = ops.resize_with_antialiasing(x, tuple(int(v) for v in g[name + "_size"]))
    assert tuple(y.shape) == g[name + "_y"].shape
    err = float((y.cpu() - torch.from_numpy(g[name + "_y"])).abs().max())
    print(f"resize {name}: max abs err {err:.2e}")
    assert err < 2e-6, err          # fp32 both sides, same roundings in the coordinate arithmetic: summation order only (measured 4e-7)


def test_pipeline_call_runs_the_pre_loop_stages_like_the_reference(dev, golden):
    """__call__ with an image_encoder and a vae (the stand-ins of tests/golden/make_golden.py): _encode_image (HIP resize,
    no CLIP normalisation, zeros for the CFG-negative half) and the noise-augmented VAE encode (mode, unscaled, zeros in
    front) hand the loop exactly what the reference's own __call__ handed it (loop.npz: clip_embed / vae_mode)."""
    import types
    from tests.golden.make_golden import FakeCLIP, FakeVAE
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    g = golden("loop")
    micro = dict(block_out_channels=(32, 32, 64, 64), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16,
                 addition_time_embed_dim=8, projection_class_embeddings_input_dim=24, layers_per_block=2, num_frames=4, in_channels=8)
    stub = types.SimpleNamespace(config=types.SimpleNamespace(**micro), device=dev)
    vae, clip = FakeVAE(), FakeCLIP(16)

    class HostCLIP:                                          # the stand-in is a CPU module: hand it the resized image on the host
        dtype = torch.float32

        def __call__(self, x):
            return clip(x.cpu())
    pipe = StableVideoDiffusionPipelineControlNet(vae=vae, image_encoder=HostCLIP(), unet=stub, controlnet=stub,
                                                  scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    seen = {}

    def fake_denoise(lat, image_latents, image_embeddings, cond, *a, **k):
        seen.update(lat=lat, image_latents=image_latents, image_embeddings=image_embeddings, cond=cond)
        return lat
    pipe.denoise = fake_denoise
    image, cond = torch.from_numpy(g["image"]), torch.from_numpy(g["cond"])
    pipe(image, controlnet_condition=cond, height=64, width=64, num_frames=4, num_inference_steps=2, min_guidance_scale=1.0,
         max_guidance_scale=3.0, fps=9, motion_bucket_id=33, noise_aug_strength=0.05, generator=torch.Generator().manual_seed(9),
         latents=torch.from_numpy(g["latents"]).clone(), output_type="latent", return_dict=False, controlnet_cond_scale=0.8)
    emb, lat = seen["image_embeddings"].float().cpu(), seen["image_latents"].float().cpu()
    assert tuple(emb.shape) == (2, 1, 16) and float(emb[0].abs().max()) == 0.0
    assert float((emb[1, 0] - torch.from_numpy(g["base_n2_clip_embed"])[0]).abs().max()) < 1e-4
    assert tuple(lat.shape) == (2, 4, 8, 8) and float(lat[0].abs().max()) == 0.0
    assert float((lat[1] - torch.from_numpy(g["base_n2_vae_mode"])[0]).abs().max()) < 1e-5
    assert tuple(seen["cond"].shape) == (2, 4, 3, 64, 64)


def test_networks_with_the_in_tree_default_head_layout(dev):
    """num_attention_heads = (5, 10, 10, 20) is the reference's constructor default (models/controlnet_sdv.py:262,
    unet...:93): head_dim 128 at level 2 (spatial attention through pt_attn_f16, temporal through the HDIM = 128 instance of
    pt_attn_temporal_f16).  Scaled down here to channels (64, 128, 256, 256) with heads (1, 2, 2, 4) = head_dims 64, 64, 128, 64."""
    from tests import parity as P
    cfg = dict(P.TINY, num_attention_heads=(1, 2, 2, 4))
    cn_o, unet_o = P.build_oracle_nets(3, cfg=cfg)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev, cfg=cfg)
    d = P.net_ladder(device=dev, latent_hw=(16, 16), seed=3, modes=("fp32",), cfg=cfg, nets=(cn_o, unet_o, cn_h, unet_h))
    print("in-tree head layout:", d)
    assert d["unet"]["hip|fp32"] < 1.0e-3 and d["controlnet_mid"]["hip|fp32"] < 1.55e-3


def test_pipeline_upcasts_a_foreign_fp16_vae_around_encode_like_the_reference(dev, golden):
    """`needs_upcasting = vae.dtype == fp16 and vae.config.force_upcast` (pipeline...:454-463): a torch-module VAE in
    fp16 is moved to fp32 for encode() and back afterwards, and sees an fp32 image; without force_upcast it sees an image in its
    own dtype.  (This package's own VAE ignores the request: its kernels accumulate in fp32 whatever they store.)"""
    import types
    from tests.golden.make_golden import FakeCLIP
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    g = golden("loop")
    micro = dict(block_out_channels=(32, 32, 64, 64), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16,
                 addition_time_embed_dim=8, projection_class_embeddings_input_dim=24, layers_per_block=2, num_frames=4, in_channels=8)
    stub = types.SimpleNamespace(config=types.SimpleNamespace(**micro), device=dev)
    clip = FakeCLIP(16)

    class HostCLIP:
        dtype = torch.float32

        def __call__(self, x):
            return clip(x.cpu())

    class HalfVAE(torch.nn.Module):
        def __init__(self, force_upcast):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1, dtype=torch.float16))
            self.config = types.SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=force_upcast)
            self.log = []

        @property
        def dtype(self):
            return self.p.dtype

        def to(self, *a, **k):
            self.log.append(("to", k.get("dtype")))
            return super().to(*a, **k)

        def encode(self, image):
            self.log.append(("encode", image.dtype, self.p.dtype))
            lat = torch.nn.functional.avg_pool2d(image.float(), 8)
            lat = torch.cat([lat, lat.mean(1, keepdim=True)], dim=1)
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: lat))
    for force in (True, False):
        vae = HalfVAE(force)
        pipe = StableVideoDiffusionPipelineControlNet(vae=vae, image_encoder=HostCLIP(), unet=stub, controlnet=stub,
                                                      scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
        pipe.denoise = lambda lat, *a, **k: lat
        pipe(torch.from_numpy(g["image"]), controlnet_condition=torch.from_numpy(g["cond"]), height=64, width=64, num_frames=4,
             num_inference_steps=2, generator=torch.Generator().manual_seed(9), output_type="latent")      # no `latents`: a CPU generator draws them
        if force:
            assert vae.log == [("to", torch.float32), ("encode", torch.float32, torch.float32), ("to", torch.float16)], vae.log
        else:
            assert vae.log == [("encode", torch.float16, torch.float16)], vae.log


@pytest.mark.parametrize("hw", [(8, 8), (8, 24)])
def test_split_cfg_halves_equal_the_full_batch(dev, hw):
    """tools/variants/split_cfg.py (measured and dropped; it keeps the product's one-CFG-half forward, ``half=``, covered): the two
    CFG halves as independent network evaluations on two streams.  Same arithmetic on half
    the rows (the temporal cross-attention's batch-interleaved context index is kept through the two-row table: the 1 x 3 tokens
    of level 3 at an 8 x 24 latent exercise the swapped table); only launch geometry differs (tile choice, split-K, GroupNorm slab sizes), so the result
    equals the full-batch loop up to fp32 summation order."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    from tools.variants.split_cfg import networks_split
    cn_o, unet_o = P.build_oracle_nets(seed=8)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    g = torch.Generator().manual_seed(5)
    h, w = hw
    lat = (torch.randn(1, 14, 4, h, w, generator=g) * 700).to(dev)
    mode = torch.randn(1, 4, h, w, generator=g).half()
    il = torch.cat([torch.zeros_like(mode), mode]).to(dev)
    e = torch.randn(1, 1, 64, generator=g).half()
    emb = torch.cat([torch.zeros_like(e), e]).to(dev)
    c1 = (torch.rand(1, 14, 3, h * 8, w * 8, generator=g) * 2 - 1).half()
    cond = torch.cat([c1, c1]).to(dev)
    full = pipe.denoise(lat, il, emb, cond, num_inference_steps=3)
    for ug in (False, True):
        split = pipe.denoise(lat, il, emb, cond, num_inference_steps=3, _networks=networks_split, use_graph=ug)
        r = P.rel_l2(split, full)
        print(f"split_cfg (graph={ug}) vs full batch at {hw}: rel-L2 {r:.2e}")
        assert r < 2e-4, r


def test_latent_sizes_the_up_path_cannot_match_are_refused(dev):
    """A latent whose height / width is not a multiple of 8 makes the reference fail in torch.cat([hidden, skip]) of the first
    up block; the MI355X path folds that concatenation into a gather, so it checks the shapes itself instead of reading out of
    bounds."""
    from tests import parity as P
    cn_o, unet_o = P.build_oracle_nets(seed=8)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, dev)
    i = {k: v.to(dev) for k, v in P.tiny_inputs(seed=1, h=5, w=9).items()}
    down, mid = cn_h(i["sample"].half(), i["t"], i["ehs"].half(), i["ids"], controlnet_cond=i["cond"].half(), return_dict=False)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        unet_h(i["sample"].half(), i["t"], i["ehs"].half(), down, mid, return_dict=False, added_time_ids=i["ids"])
