"""``pipeline/pipeline_stable_video_diffusion_controlnet_cam.py`` of the reference: the same pipeline with ``camera_cond`` as the
THIRD positional argument of ``__call__`` (``..._cam.py:316-340``; the camera inference script calls
``pipeline(image, maps[:14], cam_parameter[:14], decode_chunk_size=8, ...)``,
``infer/run_inference_vipseg_json_cam_concat_repro.py:496``) and mandatory in effect (``torch.tensor(camera_cond)`` at
``:505``).  Use it with ``posetraj_amd.controlnet_sdv_cam_infer.ControlNetSDVModel``."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from .pipeline_stable_video_diffusion_controlnet import (StableVideoDiffusionPipelineControlNet as _Base,
                                                         StableVideoDiffusionPipelineOutput, tensor2vid)

__all__ = ["StableVideoDiffusionPipelineControlNet", "StableVideoDiffusionPipelineOutput", "tensor2vid"]


class StableVideoDiffusionPipelineControlNet(_Base):
    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition: torch.FloatTensor = None, camera_cond=None, height: int = 576,
                 width: int = 1024, num_frames: Optional[int] = None, num_inference_steps: int = 25,
                 min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
                 noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None,
                 num_videos_per_prompt: Optional[int] = 1, generator=None, latents: Optional[torch.FloatTensor] = None,
                 output_type: Optional[str] = "pil", callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_dict: bool = True,
                 controlnet_cond_scale=1.0, batch_size=1, **kw):
        cam = torch.tensor(camera_cond, dtype=torch.float32)        # ..._cam.py:505: raises for None exactly like the reference
        return _Base.__call__(self, image, controlnet_condition, height, width, num_frames, num_inference_steps, min_guidance_scale,
                              max_guidance_scale, fps, motion_bucket_id, noise_aug_strength, decode_chunk_size, num_videos_per_prompt,
                              generator, latents, output_type, callback_on_step_end, callback_on_step_end_tensor_inputs, return_dict,
                              controlnet_cond_scale, batch_size, camera_cond=cam, **kw)
