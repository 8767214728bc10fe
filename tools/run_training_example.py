#!/usr/bin/env python3
"""The flow of the reference's training script (scripts/train_svd_traj_VIPSeg_14.py:935-1076 set-up, :1264-1425 loop body,
:1440-1470 checkpoint) on this package, on synthetic clips: VAE-encode the frames (tensor_to_vae_latent, :495-503), CLIP-embed the
first frame (encode_image), rasterise the trajectory maps, run ControlNetTrainer.step for a few iterations, save the ControlNet
with save_pretrained, load it back into the inference class and run one denoising call with it; with --checkpointing-steps the
loop writes checkpoint-<global_step> folders (accelerator.save_state), rotates them (--checkpoints-total-limit) and a second
invocation with --resume-from-checkpoint latest continues where the first stopped (:1224-1247).

    python tools/run_training_example.py [--steps 6] [--height 320 --width 576] [--tiny] [--out gpurun_out/controlnet_trained]
        [--svd-dir <stable-video-diffusion-img2vid dir> [--controlnet-dir <dir with controlnet/>]]
Without --svd-dir the models are random-init (no checkpoint can be fetched here): a run of the whole training path on the MI355X,
not a training result."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from posetraj_amd import (AutoencoderKLTemporalDecoder, CLIPVisionModelWithProjection, ControlNetSDVModel, EulerDiscreteScheduler,
                          SVD_SCHEDULER_CONFIG, StableVideoDiffusionPipelineControlNet, UNetSpatioTemporalConditionControlNetModel)
from posetraj_amd.training import ControlNetTrainer
from posetraj_amd.trajectory import trajectory_maps

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6); ap.add_argument("--height", type=int, default=320); ap.add_argument("--width", type=int, default=576)
ap.add_argument("--frames", type=int, default=14); ap.add_argument("--accumulation", type=int, default=2)
ap.add_argument("--tiny", action="store_true"); ap.add_argument("--out", default="gpurun_out/controlnet_trained")
ap.add_argument("--svd-dir"); ap.add_argument("--controlnet-dir")
ap.add_argument("--lr-scheduler", default="constant"); ap.add_argument("--lr-warmup-steps", type=int, default=500)
ap.add_argument("--checkpointing-steps", type=int, default=0); ap.add_argument("--checkpoints-total-limit", type=int)
ap.add_argument("--resume-from-checkpoint")
ap.add_argument("--graph", action="store_true", help="ControlNetTrainer(use_graph=True): the step replayed as a hipGraph (needs --accumulation 1)")
a = ap.parse_args()
from posetraj_amd import train_state
dev = torch.device("cuda:0")
svd, vae_cfg, clip_cfg, ce = dict(bench.SVD), dict(bench.SVD_VAE), dict(bench.CLIP_VIT_H), (16, 32, 96, 256)
if a.tiny:
    svd = dict(block_out_channels=(64, 64, 128, 128), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=1024, addition_time_embed_dim=8,
               projection_class_embeddings_input_dim=24, layers_per_block=1, num_frames=a.frames)           # (VAE and CLIP stay full-size)
    ce = (8, 8, 16, 32)
t0 = time.time()
if a.svd_dir:                                                                                                    # :900-938
    unet = UNetSpatioTemporalConditionControlNetModel.from_pretrained(a.svd_dir, subfolder="unet", device=dev, variant="fp16", keep_source=True)
    vae = AutoencoderKLTemporalDecoder.from_pretrained(a.svd_dir, subfolder="vae", device=dev, variant="fp16")
    clip = CLIPVisionModelWithProjection.from_pretrained(a.svd_dir, subfolder="image_encoder", device=dev, variant="fp16")
    controlnet = (ControlNetSDVModel.from_pretrained(a.controlnet_dir, subfolder="controlnet", device=dev, keep_source=True) if a.controlnet_dir
                  else ControlNetSDVModel.from_unet(unet))
else:
    unet = UNetSpatioTemporalConditionControlNetModel(**svd).init_random_(seed=1, device=dev, keep_source=True)    # frozen (:953)
    vae = AutoencoderKLTemporalDecoder(**vae_cfg).init_random_(seed=3, device=dev)
    clip = CLIPVisionModelWithProjection(**clip_cfg).init_random_(seed=4, device=dev)
    controlnet = ControlNetSDVModel.from_unet(unet, conditioning_embedding_out_channels=ce)                      # :935-938
max_train_steps = max(1, a.steps // a.accumulation)
trainer = ControlNetTrainer(controlnet.config, controlnet.state_dict(), unet, learning_rate=1e-5, freeze_gc=True, gradient_accumulation_steps=a.accumulation, use_graph=a.graph,
                            conditioning_dropout_prob=0.1, scaling_factor=vae.config.scaling_factor,
                            lr_scheduler=train_state.get_scheduler(a.lr_scheduler, a.lr_warmup_steps, max_train_steps, lr_init=1e-5))    # :1109-1114
global_step, first_it = 0, 0
resume = train_state.resolve_resume(a.out, a.resume_from_checkpoint)                                              # :1224-1247
if resume:
    trainer.load_state(resume)
    global_step, _, first_it = train_state.resume_position(resume, a.accumulation, max_train_steps)
    print(f"resuming from {resume}: global step {global_step}, skipping {first_it} batches; loss scale {trainer.loss_scale:g}")
pipe = StableVideoDiffusionPipelineControlNet(vae=vae, image_encoder=clip, unet=unet, controlnet=controlnet, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
print(f"set-up {time.time() - t0:.1f} s; {trainer.params.numel / 1e6:.1f} M trainable parameters")
g = torch.Generator().manual_seed(0)
for it in range(a.steps):
    if it < first_it:                                        # the reference skips the batches a resumed epoch has already seen (:1259)
        torch.rand(1, a.frames, 3, a.height, a.width, generator=g)
        continue
    # a synthetic "batch": pixel_values [1, F, 3, H, W] in [-1, 1], tracks -> trajectory maps, motion value (:1268-1280)
    pixel_values = (torch.rand(1, a.frames, 3, a.height, a.width, generator=g) * 2 - 1).to(dev)
    tracks = bench.synth_tracks(a.frames, a.height, a.width, it)
    maps = trajectory_maps(tracks, [a.height, a.width], (a.height, a.width, 3), num_frames=a.frames, device=dev)      # PIL-free [F, 3, H, W] in [-1, 1]
    t1 = time.time()
    latents = vae.encode(pixel_values[0]).latent_dist.sample().unsqueeze(0) * vae.config.scaling_factor            # tensor_to_vae_latent
    emb = pipe._encode_image(pixel_values[:, 0].float().add(1).div(2), dev, 1, False)                                           # encode_image(pixel_values[:, 0])
    out = trainer.step(latents, emb, torch.tensor([127.0]), maps.unsqueeze(0))
    torch.cuda.synchronize()
    print(f"iteration {it}: loss {out['loss']:.4f} (spatial {out['loss_spatial']:.4f}) grad norm {('%.3e' % out['grad_norm']) if 'grad_norm' in out else '(accumulating)'} "
          f"optimizer stepped: {out['stepped']} lr {trainer.last_lr:.2e}  [{(time.time() - t1) * 1e3:.0f} ms incl. VAE encode + CLIP]")
    if out["stepped"] is not None:                           # accelerator.sync_gradients (:1428-1467)
        global_step += 1
        if a.checkpointing_steps and global_step % a.checkpointing_steps == 0:
            gone = train_state.rotate_checkpoints(a.out, a.checkpoints_total_limit)
            t2 = time.time()
            trainer.save_state(os.path.join(a.out, f"checkpoint-{global_step}"))
            print(f"  saved state to {a.out}/checkpoint-{global_step} in {time.time() - t2:.1f} s" + (f" (removed {', '.join(gone)})" if gone else ""))
# checkpoint (:1440-1470) and back into the inference class
trained = ControlNetSDVModel(**{k: v for k, v in dict(controlnet.config).items() if not k.startswith("_")}).load_state_dict(trainer.state_dict(), dev, keep_source=True)
trained.save_pretrained(os.path.join(a.out, "controlnet"))
again = ControlNetSDVModel.from_pretrained(a.out, subfolder="controlnet", device=dev)
pipe.controlnet = again
frames = pipe(pixel_values[0, 0].add(1).div(2).unsqueeze(0).cpu(), maps, height=a.height, width=a.width, num_frames=a.frames, decode_chunk_size=8,
              num_inference_steps=3, output_type="pt").frames
print(f"saved and re-loaded {a.out}/controlnet ({sum(v.numel() for v in trainer.state_dict().values()) / 1e6:.1f} M parameters); "
      f"3-step sample with the trained ControlNet: {len(frames)} clip of frames {tuple(frames[0].shape)}, finite {bool(torch.isfinite(frames[0]).all())}")
