#!/usr/bin/env python3
"""The headline parity claim at the headline geometry (VERDICT r04 #2a): the full-width ControlNet + U-Net (1.52 B + 0.68 B
parameters, seeded random init), BASELINE configs[2] = 14 x 576 x 1024 (latent 72 x 128), CFG, the WHOLE 25-step Euler/Karras
loop of the pipeline (hipGraph + two streams, as bench.py and __call__ run it) on the MI355X against the fp32 CPU oracle.
The oracle's 25 iterations are 25 x 123 TFLOP of fp32 on host cores (hours), so the two sides run in two places:

    gpurun:           python tools/full_width_L_25step_parity.py --hip   gpurun_out/L25_hip.pt      (seconds on the MI355X)
    build container:  python tools/full_width_L_25step_parity.py --oracle gpurun_out/L25_hip.pt --state /tmp/L25_oracle.pt

Weights and inputs are rebuilt on both sides from seeds (oracle/init.py, a CPU generator); the dump carries checksums of
them so that a mismatch between the two machines is caught, and the HIP latents after steps 1, 5 and 25.  The oracle side
checkpoints after every iteration (--state) and resumes.  Output committed under profiles/r05/.

    build container:  python tools/full_width_L_25step_parity.py --export tests/golden/loop_L_25step_oracle.npz --state /tmp/L25_oracle.pt

writes the ORACLE side of a finished run (the fp32 final latents, the checksums of the seeded weights / inputs they belong to
and the distances the run measured) as the fixture tests/test_parity_ladder_gpu.py::test_config2_full_width_25_step_loop_
against_the_stored_oracle_latents asserts against - the HIP side is re-run by that test every time (round 6)."""
import argparse, os, resource, sys, time
sys.path.insert(0, os.getcwd())
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--hip", metavar="OUT.pt"); ap.add_argument("--oracle", metavar="HIP.pt"); ap.add_argument("--state", default="/tmp/L25_oracle.pt")
ap.add_argument("--export", metavar="OUT.npz")
ap.add_argument("--camera", action="store_true", help="the camera twin (controlnet_sdv_cam, BASELINE configs[4]): camera ControlNet (seed 23) + per-frame R|T")
ap.add_argument("--steps", type=int, default=25); ap.add_argument("--latent", type=int, nargs=2, default=(72, 128))
ap.add_argument("--threads", type=int, default=0); ap.add_argument("--seed", type=int, default=13)
a = ap.parse_args()
from tests import parity as P
from oracle import loop as OL, sched as OS
if a.threads:
    torch.set_num_threads(a.threads)
F, (h, w) = 14, a.latent


inputs = lambda xdim: P.loop_inputs(a.seed, F, h, w, xdim)
digest, weights_digest = P.tensor_digest, P.weights_digest


if a.export:                                                           # a finished oracle run -> the committed fixture (no network is rebuilt)
    import numpy as np
    s = torch.load(a.state)
    assert s["i"] == s["sig"]["steps"] == a.steps, "the oracle run behind --state has not finished"
    np.savez_compressed(a.export, latents=s["latents"].numpy().astype(np.float32), inputs_sha=np.array(s["sig"]["inputs"]),
                        weights_sha=np.array(s["sig"]["weights"]), steps=np.array(a.steps), latent_hw=np.array(s["sig"]["latent"]),
                        net_seed=np.array(7), input_seed=np.array(a.seed), controlnet_cond_scale=np.array(0.9), camera=np.array(int(bool(s["sig"].get("camera")))),
                        rel_l2_measured=np.array([s["rel"][k] for k in sorted(s["rel"])]), rel_l2_after=np.array(sorted(s["rel"])),
                        oracle_seconds=np.array(s["secs"]))
    print(f"wrote {a.export}: {os.path.getsize(a.export) / 1e6:.2f} MB, sig {s['sig']}")
    sys.exit(0)

t0 = time.time()
cn_o, unet_o = P.build_oracle_nets(7, cfg=P.SVD_CFG, ce=P.SVD_CE)
cam = None
if a.camera:
    cn_o = P.build_oracle_camera_controlnet()
    cam = P.loop_camera_input(a.seed, F)
lat, il, emb, cond = inputs(unet_o.config.cross_attention_dim)
so = OS.OracleEulerDiscreteScheduler(**OS.SVD_SCHEDULER_CONFIG); so.set_timesteps(a.steps)
lat0 = lat * so.init_noise_sigma
sig = dict(inputs=digest(lat0, il, emb, cond, *([cam] if cam is not None else [])), weights=weights_digest(cn_o, unet_o), steps=a.steps, latent=(h, w))
if a.camera:
    sig["camera"] = True
print(f"build {time.time() - t0:.1f} s   {sig}", flush=True)

if a.hip:
    from posetraj_amd import EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet, SVD_SCHEDULER_CONFIG
    cn_h, unet_h = P.build_hip_nets(cn_o, unet_o, "cuda:0", camera=a.camera, cfg=P.SVD_CFG, ce=P.SVD_CE)
    out, want = {}, {1, 5, a.steps}
    def grab(pipe_, i, t, kw):                                         # host-side only: the loop's launches are the bench's
        if i + 1 in want:
            out[i + 1] = kw["latents"].detach().float().cpu().clone()
        return {}
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet_h, controlnet=cn_h, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
    t = time.time()
    o = pipe.denoise(lat0.cuda(), il.cuda(), emb.cuda(), cond.cuda(), num_inference_steps=a.steps, controlnet_cond_scale=0.9,
                     use_graph=True, overlap_streams=True, callback_on_step_end=grab, camera_cond=None if cam is None else cam.cuda())
    torch.cuda.synchronize()
    assert torch.equal(o.float().cpu(), out[a.steps])
    print(f"HIP path: {a.steps} iterations in {time.time() - t:.1f} s (graph capture included)", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.hip)), exist_ok=True)
    torch.save(dict(sig=sig, hip=out, device=torch.cuda.get_device_name(0)), a.hip)
    print(f"wrote {a.hip}")
    sys.exit(0)

d = torch.load(a.oracle)
assert d["sig"] == sig, f"the two sides built different weights / inputs:\n  hip    {d['sig']}\n  oracle {sig}"
g = OL.guidance_ramp(1.0, 3.0, F, 1, lat0.dtype, lat0.ndim)
ids = OL.hot_added_time_ids(emb.dtype)
il5 = il.unsqueeze(1).repeat(1, F, 1, 1, 1)
state = dict(i=0, latents=lat0, secs=[])
if os.path.exists(a.state):
    s = torch.load(a.state)
    if s.get("sig") == sig:
        state = s
        print(f"resuming after iteration {state['i']}", flush=True)
latents = state["latents"]
so._step_index = None
with torch.no_grad():
    for i, t in enumerate(so.timesteps):
        if i < state["i"]:
            so._step_index = i + 1                                    # the scheduler's only per-step state
            continue
        t1 = time.time()
        x = so.scale_model_input(torch.cat([latents] * 2), t)
        if so._step_index is None:
            so._init_step_index(t)
        x = torch.cat([x, il5], dim=2)
        down, mid = cn_o(x, t, encoder_hidden_states=emb, controlnet_cond=cond, added_time_ids=ids, conditioning_scale=0.9,
                         guess_mode=False, return_dict=False, **(dict(camera_cond=cam) if cam is not None else {}))
        pred = unet_o(x, t, encoder_hidden_states=emb, down_block_additional_residuals=down, mid_block_additional_residual=mid,
                      added_time_ids=ids, return_dict=False)[0]
        un, co = pred.chunk(2)
        latents = so.step(un + g * (co - un), t, latents).prev_sample
        state["secs"].append(time.time() - t1)
        state.update(i=i + 1, latents=latents, sig=sig)
        line = f"oracle iteration {i + 1}/{a.steps}: {state['secs'][-1]:.0f} s"
        if (i + 1) in d["hip"]:
            r = P.rel_l2(d["hip"][i + 1], latents)
            state.setdefault("rel", {})[i + 1] = r
            line += f"   rel-L2 of the HIP latents after {i + 1} iterations vs the fp32 oracle = {r:.3e}"
        print(line, flush=True)
        torch.save(state, a.state)
print(f"full-width networks, {a.steps}-step CFG loop at the {h} x {w} latent (14 x {h * 8} x {w * 8}) on {d['device']}: "
      f"rel-L2 of the final latents vs the fp32 oracle = {state['rel'][a.steps]:.3e}   (after 1 / 5 iterations: "
      f"{state['rel'].get(1, float('nan')):.3e} / {state['rel'].get(5, float('nan')):.3e}; oracle {sum(state['secs']) / 60:.0f} min on "
      f"{torch.get_num_threads()} host threads, peak {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB)", flush=True)
