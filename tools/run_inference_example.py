#!/usr/bin/env python3
"""The flow of the reference's inference script (scripts/run_inference_vipseg_json_repro.py:335-339,420-457) on this package:
load (or random-init) the models, build the 13 + 1 trajectory maps from a tracks JSON on the device, call the pipeline the way the
script does, save the frames as a GIF.

    python tools/run_inference_example.py --image first_frame.png --tracks tracks.json --out out.gif \\
        [--svd-dir <stable-video-diffusion-img2vid dir> --controlnet-dir <dir with controlnet/>] [--height 320 --width 576]
Without checkpoint directories the models are random-init at full SVD size (the output is then noise - a smoke run of the whole
image-to-video path on the MI355X)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import PIL.Image
import torch
import bench
from posetraj_amd import (AutoencoderKLTemporalDecoder, CLIPVisionModelWithProjection, ControlNetSDVModel, EulerDiscreteScheduler,
                          SVD_SCHEDULER_CONFIG, StableVideoDiffusionPipelineControlNet, UNetSpatioTemporalConditionControlNetModel)
from posetraj_amd.trajectory import load_tracks, trajectory_maps

ap = argparse.ArgumentParser()
ap.add_argument("--image"); ap.add_argument("--tracks"); ap.add_argument("--out", default="gpurun_out/example.gif")
ap.add_argument("--svd-dir"); ap.add_argument("--controlnet-dir")
ap.add_argument("--height", type=int, default=320); ap.add_argument("--width", type=int, default=576)
ap.add_argument("--steps", type=int, default=25)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.svd_dir and a.controlnet_dir:                                   # scripts/...:335-339
    controlnet = ControlNetSDVModel.from_pretrained(a.controlnet_dir, subfolder="controlnet", device=dev)
    unet = UNetSpatioTemporalConditionControlNetModel.from_pretrained(a.svd_dir, subfolder="unet", device=dev, variant="fp16")
    pipe = StableVideoDiffusionPipelineControlNet.from_pretrained(a.svd_dir, controlnet=controlnet, unet=unet, device=dev, variant="fp16")
else:
    unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=1, device=dev)
    controlnet = ControlNetSDVModel(**bench.SVD).init_random_(seed=2, device=dev)
    pipe = StableVideoDiffusionPipelineControlNet(vae=AutoencoderKLTemporalDecoder(**bench.SVD_VAE).init_random_(seed=3, device=dev),
                                                  image_encoder=CLIPVisionModelWithProjection(**bench.CLIP_VIT_H).init_random_(seed=4, device=dev),
                                                  unet=unet, controlnet=controlnet, scheduler=EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG))
if a.image:
    image = PIL.Image.open(a.image).convert("RGB")
else:
    image = PIL.Image.fromarray(np.random.default_rng(0).integers(0, 256, (a.height, a.width, 3), dtype=np.uint8))
original_size = np.array(image).shape                                # scripts/...:420
tracks = load_tracks(a.tracks) if a.tracks else bench.synth_tracks(14, original_size[0], original_size[1], 0)
maps = trajectory_maps(tracks, [a.height, a.width], original_size, num_frames=14, device=dev)        # :426-447 without cv2
t0 = time.time()
frames = pipe(image, maps, decode_chunk_size=8, num_frames=14, motion_bucket_id=10, controlnet_cond_scale=1.0, width=a.width,
              height=a.height, num_inference_steps=a.steps).frames   # :451
torch.cuda.synchronize()
print(f"{len(frames[0])} frames of {frames[0][0].size} in {time.time() - t0:.2f} s (first call: includes the hipGraph capture)")
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
frames[0][0].save(a.out, format="GIF", append_images=frames[0][1:], save_all=True, duration=200, loop=0)
print("wrote", a.out)
