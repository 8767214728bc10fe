"""The parity ladder on the MI355X: the HIP path against the oracle run at three storage precisions
(oracle/quant.py) on identical inputs, so that dtype error and implementation error are asserted separately:

    HIP <-> fp16-fused   implementation only: the oracle rounds exactly where the HIP path stores fp16
    fp16-fused <-> fp32  what fp16 storage of the fused graph costs (pure dtype, no HIP code involved)
    HIP <-> fp32         the number the north star quotes (<= 1e-3 for one network forward)

plus the BASELINE.json configurations no other -m gpu test reaches: configs[0] at its stated 64 x 64 latent,
configs[2] / configs[4] (14 x 576 x 1024: latent 72 x 128, with and without the camera branch) with the tiny nets, and
one full-width level-0 layer pair at the 72 x 128 geometry."""
import pytest
import torch

from tests import parity as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Measured on MI355X (profiles/r02/parity_ladder_v2.txt) with the residual stream stored as fp16 pairs (ops.WIDE_STREAM:
# shortcut, spatial-resnet and resblock outputs; r02 before that: U-Net 1.14e-3, ControlNet mid 1.54e-3, loop 2.25e-3).
# What remains is the fp16 STORAGE noise of the other tensors, not of the kernels: the oracle itself, run with fp32
# arithmetic but rounding exactly where the HIP path stores ("fp16-fused"), sits at the same distance from the fp32
# result (U-Net 7.0e-4 / HIP 7.8e-4, ControlNet mid block 1.22e-3 / 1.24e-3, one CFG loop iteration 1.80e-3 / 1.76e-3 at
# 16 x 16 and 1.45e-3 / 1.42e-3 at 40 x 72), the reference's own every-op fp16 execution ("fp16") is further away
# (1.25e-3 / 1.8e-3 / 2.8e-3), and every kernel alone is exact up to its output rounding (tools/op_ladder.py: impl <= 2e-5
# for GEMM / conv / norm, <= 3e-4 for attention).  Asserted: the north star's 1e-3 for the U-Net forward, measured x 1.25
# elsewhere, and "not worse than the storage model / than the reference in fp16".
TOL_UNET_FP32 = 1.0e-3       # HIP <-> fp32 oracle, U-Net forward            (measured 7.8e-4 .. 8.0e-4)
TOL_CN_FP32 = 1.55e-3        # HIP <-> fp32 oracle, ControlNet mid residual  (measured 1.21e-3 .. 1.24e-3, 60+ blocks deep)
TOL_LOOP_FP32 = 2.2e-3       # HIP <-> fp32 oracle, one loop iteration       (measured 1.42e-3 .. 1.76e-3: CFG amplifies)


# full-width layer pairs at levels 1-3 (profiles/r02/full_width_levels.txt; unchanged in profiles/r05/parity_ladder_prints_r05.txt):
# measured x 1.25 (round 5; x 1.5 before)
TOL_FULL_RES = {1: 4.05e-4, 2: 4.1e-4, 3: 3.8e-4}    # measured 3.23e-4, 3.27e-4, 3.04e-4
TOL_FULL_ATT = {1: 7.35e-4, 2: 6.75e-4, 3: 7.1e-4}   # measured 5.86e-4, 5.39e-4, 5.66e-4


# whole networks at the full SVD width, 16 x 16 latent (profiles/r02/full_width_levels.txt): the north star's 1e-3 for the
# U-Net (measured 6.5e-4; its fp16-fused storage model 5.9e-4), measured x 1.25 for the ControlNet mid residual (1.09e-3; storage model 1.11e-3)
TOL_FULL_UNET, TOL_FULL_CN = 1.0e-3, 1.36e-3
TOL_FULL_LOOP = 1.42e-3       # one CFG loop iteration on the full-width networks, 16 x 16 latent: measured 1.13e-3 (x 1.25)
TOL_BLOCKS_FIXTURE = 1.0e-3   # HIP blocks <-> reference-run blocks.npz (fp32): 1-3 layer pairs deep
TOL_FULL_LOOP_L = 1.77e-3     # ... and at configs[2]'s 72 x 128 latent, the benched workload: measured 1.41e-3 (x 1.25, the stated tolerance; 1.47e-3 before round 5)
TOL_FULL_LOOP_M = 1.91e-3     # the same at BASELINE configs[1]'s 40 x 72 latent: measured 1.53e-3 (x 1.25; 1.58e-3 before round 5; tiny nets there: 1.42e-3)


def test_network_ladder():
    d = P.net_ladder(DEV, latent_hw=(16, 16))
    for net, tol in (("controlnet_mid", TOL_CN_FP32), ("unet", TOL_UNET_FP32)):
        assert d[net]["hip|fp32"] < tol, (net, d[net])
        # no further from the exact result than its own storage model and than the reference's fp16 execution
        assert d[net]["hip|fp32"] < 1.15 * d[net]["fp16-fused|fp32"], (net, d[net])
        assert d[net]["hip|fp32"] < 1.05 * d[net]["fp16|fp32"], (net, d[net])
        # the two fp16 oracles and the HIP path are three independent realisations of the same rounding noise
        assert d[net]["hip|fp16-fused"] < 1.6 * d[net]["fp16-fused|fp32"], (net, d[net])


def test_one_loop_iteration_ladder():
    r, out, ref, d = P.run_tiny_pipeline_parity(steps=1, latent_hw=(16, 16), device=DEV, return_all=True,
                                                modes=("fp32", "fp16-fused", "fp16"))
    assert d["hip|fp32"] < TOL_LOOP_FP32, d
    assert d["hip|fp32"] < 1.15 * d["fp16-fused|fp32"], d
    assert d["hip|fp32"] < 1.05 * d["fp16|fp32"], d


def test_whole_25_step_sampler_loop():
    """The complete Euler loop of BASELINE's configurations (25 steps, CFG, Karras sigmas 700 -> 0.002) on the tiny nets:
    the error of one iteration (1.4e-3 ... 1.8e-3) does not accumulate over the trajectory - the final latents are closer
    to the fp32 oracle's than a single step's (measured 7.0e-4 at 16 x 16, 5.8e-4 at 8 x 8; 5 steps: 1.4e-3)."""
    r, out, ref, d = P.run_tiny_pipeline_parity(steps=25, latent_hw=(16, 16), device=DEV, return_all=True,
                                                modes=("fp32", "fp16-fused"), use_graph=True, overlap_streams=True)
    assert d["hip|fp32"] < 1.0e-3, d
    assert d["hip|fp32"] < 1.15 * d["fp16-fused|fp32"], d


def test_config0_tiny_nets_at_64x64_latent_two_steps():
    r = P.run_tiny_pipeline_parity(steps=2, latent_hw=(64, 64), device=DEV)
    assert r < TOL_LOOP_FP32, r


@pytest.mark.parametrize("camera", [False, True])
def test_config2_and_4_geometry_72x128_latent(camera):
    """14 x 576 x 1024 (latent 72 x 128: S = 9216 / 2304 / 576 / 144 tokens), one loop iteration, hipGraph + two streams
    as bench.py runs it; camera=True is the controlnet_sdv_cam branch of configs[4]."""
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(72, 128), device=DEV, camera=camera, use_graph=True,
                                   overlap_streams=True)
    assert r < TOL_LOOP_FP32, r


def test_full_width_level0_layer_pair_at_72x128():
    """SpatioTemporalResBlock(320 -> 320) + TransformerSpatioTemporalModel(5 x 64) at full SVD width, 14 x 72 x 128,
    CFG batch 2: every level-0 shape of the bench workload (258048-row GEMMs, S = 9216 attention) against the oracle."""
    r_res, r_att = P.full_width_level0_block(DEV)
    print(f"level 0 at 72x128: resblock {r_res:.3e}  transformer {r_att:.3e}")
    assert r_res < 4.9e-4, r_res        # measured 3.9e-4 (x 1.25)
    assert r_att < 7.5e-4, r_att        # measured 6.0e-4 with three launches around the feed-forward; the fused prologue is closer


def test_level0_layer_pair_with_and_without_the_fused_prologue():
    """The level-0 transformer with `ops.FUSED_PRE` on (out-projection + residual + LayerNorm inside the feed-forward launch, the
    default) and off (three launches) against the oracle at a 16 x 24 latent: both inside the level's bound, the fused form no further
    from fp32 than the three launches (its residual stream stays fp32 inside the kernel)."""
    from posetraj_amd import ops
    keep = ops.FUSED_PRE
    try:
        ops.FUSED_PRE = True
        _, r_on = P.full_width_level0_block(DEV, latent_hw=(16, 24))
        ops.FUSED_PRE = False
        _, r_off = P.full_width_level0_block(DEV, latent_hw=(16, 24))
    finally:
        ops.FUSED_PRE = keep
    print(f"level-0 transformer vs fp32 oracle: fused prologue {r_on:.3e}, three launches {r_off:.3e}")
    assert r_on < 6.6e-4 and r_off < 7.1e-4, (r_on, r_off)          # measured 5.25e-4 / 5.68e-4 (x 1.25)
    assert r_on < 1.05 * r_off, (r_on, r_off)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_full_width_layer_pair_at_deeper_levels(level):
    """The same layer pair at the width and geometry of levels 1-3 of the bench workload (640 ch @ 36 x 64, 1280 ch @
    18 x 32 and 9 x 16: the 64512-, 16128- and 4032-row GEMMs incl. split-K, S = 2304 / 576 / 144 attention)."""
    r_res, r_att = P.full_width_block(level, DEV)
    print(f"level {level}: resblock {r_res:.3e}  transformer {r_att:.3e}")
    assert r_res < TOL_FULL_RES[level], r_res
    assert r_att < TOL_FULL_ATT[level], r_att


@pytest.fixture(scope="module")
def full_width_nets():
    """ControlNet + U-Net at the full SVD width (1.52 B + 0.68 B parameters, seeded random init): the oracle's modules on
    the host and the HIP models built from their state dicts.  ~9 GB of host memory for the length of this module."""
    cn_o, unet_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, DEV, cfg=P.SVD_CFG, ce=P.SVD_CE)
    return cn_o, unet_o, cn_h, unet_h


def test_full_width_networks_against_the_oracle(full_width_nets):
    """The WHOLE ControlNet and U-Net at the full SVD width (all 2 x 16 + 1 resblocks / 2 x 12 transformers deep) against
    the fp32 CPU oracle, 14 frames at a 16 x 16 latent (128 x 128 px), CFG batch 2 - the configuration of BASELINE
    configs[1..4] at a geometry the oracle finishes in seconds - and one loop iteration of the pipeline on the same networks."""
    nets = full_width_nets
    d = P.net_ladder(DEV, latent_hw=(16, 16), modes=("fp32", "fp16-fused", "fp16"), seed=7, nets=nets)
    print("full-width nets:", d)
    assert d["unet"]["hip|fp32"] < TOL_FULL_UNET, d
    assert d["controlnet_mid"]["hip|fp32"] < TOL_FULL_CN, d
    for net in ("unet", "controlnet_mid"):                  # not further from the exact result than its own storage model
        assert d[net]["hip|fp32"] < 1.3 * d[net]["fp16-fused|fp32"], (net, d[net])      # measured 1.13 / 1.01
        assert d[net]["hip|fp32"] < 1.05 * d[net]["fp16|fp32"], (net, d[net])           # ... nor than the reference run in fp16
    # ... and one CFG loop iteration of the pipeline on the same networks (hipGraph + two streams, as bench.py runs it)
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(16, 16), device=DEV, nets=nets, seed=7, use_graph=True,
                                   overlap_streams=True)
    print(f"full-width loop iteration: {r:.3e}")
    assert r < TOL_FULL_LOOP, r


def test_config1_full_width_loop_iteration_at_320x576(full_width_nets):
    """BASELINE configs[1] as stated: the full-width networks, 14 x 320 x 576 (latent 40 x 72, S = 2880 / 720 / 180 / 45),
    CFG, one loop iteration of the pipeline (hipGraph + two streams) against the fp32 CPU oracle - 33 TFLOP on the host."""
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(40, 72), device=DEV, nets=full_width_nets, seed=11, use_graph=True,
                                   overlap_streams=True)
    print(f"configs[1] full-width loop iteration at 40x72: {r:.3e}")
    assert r < TOL_FULL_LOOP_M, r


def test_config2_full_width_loop_iteration_at_576x1024(full_width_nets):
    """BASELINE configs[2] - the benched workload itself: the full-width networks, 14 x 576 x 1024 (latent 72 x 128, S = 9216 /
    2304 / 576 / 144), CFG, one loop iteration of the pipeline (hipGraph + two streams, as bench.py runs it) against the fp32
    CPU oracle - 123 TFLOP on the host (~3 min on 16 cores, 22 GB).  Round 2 ran this as a tool only (1.47e-3)."""
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(72, 128), device=DEV, nets=full_width_nets, seed=13, use_graph=True,
                                   overlap_streams=True)
    print(f"configs[2] full-width loop iteration at 72x128: {r:.3e}")
    assert r < TOL_FULL_LOOP_L, r


def test_full_width_whole_25_step_loop(full_width_nets):
    """The complete 25-step Euler loop (CFG, Karras sigmas 700 -> 0.002) on the FULL-WIDTH networks at a 16 x 16 latent: the
    north star's <= 1e-3 on the pipeline output (round 2, as a tool: 4.8e-4)."""
    r = P.run_tiny_pipeline_parity(steps=25, latent_hw=(16, 16), device=DEV, nets=full_width_nets, seed=7, use_graph=True,
                                   overlap_streams=True)
    print(f"full-width 25-step loop at 16x16: {r:.3e}")
    assert r < 1.0e-3, r


@pytest.mark.parametrize("fixture", ["loop_L_25step_oracle", "loop_M_25step_oracle", "loop_M_cam_25step_oracle", "loop_L_cam_25step_oracle"])
def test_config2_full_width_25_step_loop_against_the_stored_oracle_latents(full_width_nets, golden, fixture):
    """The north star's number at the headline configuration, in the GPU suite: the full-width networks, BASELINE
    configs[2] (14 x 576 x 1024, latent 72 x 128), CFG, the WHOLE 25-step loop (hipGraph + two streams, as bench.py runs it).  Only
    the HIP side runs here (3-4 s); the fp32 oracle's final latents - 25 x 123 TFLOP, 157 min of host time - are the committed
    fixture tests/golden/loop_L_25step_oracle.npz, written by `tools/full_width_L_25step_parity.py --export`, and the fixture's
    checksums of the seeded weights and inputs must match what this process builds.  Measured 5.36e-4 (profiles/r05/)."""
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    from oracle import sched as OS
    fx = golden(fixture)                                     # L: BASELINE configs[2], 72 x 128 latent; M: configs[1], 40 x 72 (round 6, later)
    cn_o, unet_o, cn_h, unet_h = full_width_nets
    cam = None
    if "camera" in fx.files and int(fx["camera"]):          # BASELINE configs[4]: the camera twin's ControlNet (seed 23) + per-frame R|T
        from posetraj_amd.controlnet_sdv import ControlNetSDVModel
        cn_o = P.build_oracle_camera_controlnet()
        cn_h = ControlNetSDVModel(**P.SVD_CFG, conditioning_embedding_out_channels=P.SVD_CE, camera=True).load_state_dict(cn_o.state_dict(), DEV)
        cam = P.loop_camera_input(int(fx["input_seed"]), 14)
    steps, (h, w) = int(fx["steps"]), tuple(int(v) for v in fx["latent_hw"])
    assert int(fx["net_seed"]) == 7 and P.weights_digest(cn_o, unet_o) == str(fx["weights_sha"]), "fixture belongs to other weights"
    lat, il, emb, cond = P.loop_inputs(int(fx["input_seed"]), 14, h, w, unet_o.config.cross_attention_dim)
    so = OS.OracleEulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG); so.set_timesteps(steps)
    lat0 = lat * so.init_noise_sigma
    assert P.tensor_digest(lat0, il, emb, cond, *([cam] if cam is not None else [])) == str(fx["inputs_sha"]), "fixture belongs to other inputs"
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    out = pipe.denoise(lat0.to(DEV), il.to(DEV), emb.to(DEV), cond.to(DEV), num_inference_steps=steps,
                       controlnet_cond_scale=float(fx["controlnet_cond_scale"]), use_graph=True, overlap_streams=True,
                       camera_cond=None if cam is None else cam.to(DEV))
    torch.cuda.synchronize()
    r = P.rel_l2(out, torch.from_numpy(fx["latents"]))
    print(f"full-width 25-step loop at {h}x{w} vs the stored fp32 oracle latents ({fixture}): {r:.3e}")
    assert r < 1.0e-3, r


def test_config4_full_width_camera_controlnet(full_width_nets):
    """BASELINE configs[4]: controlnet_sdv_cam at the full SVD width (cc_projection 268 -> 256 on the 1/8-resolution map):
    all 12 taps + mid of the camera ControlNet forward, and one CFG loop iteration of the camera pipeline with the
    full-width U-Net, against the fp32 oracle (16 x 16 latent)."""
    import contextlib, io
    from oracle import init as OI, nets as ON
    from posetraj_amd.controlnet_sdv import ControlNetSDVModel
    _, unet_o, _, unet_h = full_width_nets
    with contextlib.redirect_stdout(io.StringIO()):
        cn_o = OI.seeded_init_(ON.ControlNetSDVModel(**P.SVD_CFG, conditioning_embedding_out_channels=P.SVD_CE, camera=True),
                               seed=23).eval()
    with torch.no_grad():
        for p_ in cn_o.parameters():
            p_.copy_(p_.half().float())
    cn_h = ControlNetSDVModel(**P.SVD_CFG, conditioning_embedding_out_channels=P.SVD_CE, camera=True).load_state_dict(
        cn_o.state_dict(), DEV)
    i = P.tiny_inputs(seed=31, h=16, w=16, xdim=unet_o.config.cross_attention_dim)
    j = {k: v.to(DEV) for k, v in i.items()}
    with torch.no_grad():
        down_o, mid_o = cn_o(i["sample"], i["t"], i["ehs"], i["ids"], controlnet_cond=i["cond"], camera_cond=i["cam"],
                             return_dict=False)
    down_h, mid_h = cn_h(j["sample"].half(), j["t"], j["ehs"].half(), j["ids"], controlnet_cond=j["cond"].half(),
                         camera_cond=j["cam"].half(), return_dict=False)
    errs = [P.rel_l2(a, b) for a, b in zip(down_h, down_o)] + [P.rel_l2(mid_h, mid_o)]
    print("full-width camera ControlNet taps:", [f"{e:.2e}" for e in errs])
    assert len(errs) == 13 and max(errs) < TOL_FULL_CN, errs
    r = P.run_tiny_pipeline_parity(steps=1, latent_hw=(16, 16), device=DEV, camera=True, nets=(cn_o, unet_o, cn_h, unet_h),
                                   seed=29, use_graph=True, overlap_streams=True)
    print(f"full-width camera loop iteration: {r:.3e}")
    assert r < TOL_FULL_LOOP, r


def test_hip_blocks_against_the_reference_run_fixture(golden):
    """tests/golden/blocks.npz holds outputs of the REFERENCE's own forwards (/root/reference/models/modified_svd.py:118-348,
    run by tests/golden/make_golden.py over the oracle's leaf modules): the HIP transformer (Q3 interleave live: CFG batch
    2), cross-attention down block (taps, downsampler) and up block (skip order, 2-source concat, upsampler)."""
    d = P.hip_blocks_vs_fixture(golden("blocks"), DEV)
    print("hip blocks vs reference-run fixture:", {k: f"{v:.2e}" for k, v in d.items()})
    for k, v in d.items():
        assert v < TOL_BLOCKS_FIXTURE, (k, d)
