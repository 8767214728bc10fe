"""posetraj_amd - MI355X-native implementation of PoseTraj's denoising hot path.

Same class names and call signatures as the reference modules it replaces:

    reference module                                             this package
    models/controlnet_sdv.py                                  -> posetraj_amd.controlnet_sdv
    models/controlnet_sdv_cam_infer.py                        -> posetraj_amd.controlnet_sdv_cam_infer
    models/unet_spatio_temporal_condition_controlnet.py       -> posetraj_amd.unet_spatio_temporal_condition_controlnet
    utils/scheduling_euler_discrete_karras_fix.py             -> posetraj_amd.scheduling_euler_discrete_karras_fix
    pipeline/pipeline_stable_video_diffusion_controlnet       -> posetraj_amd.pipeline_stable_video_diffusion_controlnet
    pipeline/pipeline_stable_video_diffusion_controlnet_cam   -> posetraj_amd.pipeline_stable_video_diffusion_controlnet_cam
    diffusers.models.AutoencoderKLTemporalDecoder (pipeline...:26) -> posetraj_amd.autoencoder_kl_temporal_decoder
    transformers.CLIPVisionModelWithProjection (pipeline...:22)   -> posetraj_amd.clip_vision
    scripts/run_inference_vipseg_json_repro.py:426-449 (maps)  -> posetraj_amd.trajectory
    scripts/train_svd_traj_VIPSeg_14.py:1264-1425 (the step)   -> posetraj_amd.training (ControlNetTrainer; tape: autodiff,
                                                                  train_graph; data-parallel exchange: grad_sync;
                                                                  lr schedule, save_state / load_state / resume: train_state)

All tensor arithmetic runs in ``libposetraj_hip.so`` (HIP, gfx950); there is no CPU / eager-PyTorch fallback.
"""
from .autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
from .clip_vision import CLIPVisionModelWithProjection
from .controlnet_sdv import ControlNetOutput, ControlNetSDVModel
from .pipeline_stable_video_diffusion_controlnet import (StableVideoDiffusionControlNetPipeline,
                                                         StableVideoDiffusionPipelineControlNet,
                                                         StableVideoDiffusionPipelineOutput)
from .training import ControlNetTrainer, controlnet_training_loss
from .scheduling_euler_discrete_karras_fix import (SVD_SCHEDULER_CONFIG, EulerDiscreteScheduler,
                                                   EulerDiscreteSchedulerOutput)
from .unet_spatio_temporal_condition_controlnet import (UNetSpatioTemporalConditionControlNetModel,
                                                        UNetSpatioTemporalConditionOutput)

__all__ = ["AutoencoderKLTemporalDecoder", "CLIPVisionModelWithProjection", "ControlNetOutput", "ControlNetSDVModel", "ControlNetTrainer",
           "controlnet_training_loss", "StableVideoDiffusionControlNetPipeline",
           "StableVideoDiffusionPipelineControlNet", "StableVideoDiffusionPipelineOutput", "SVD_SCHEDULER_CONFIG",
           "EulerDiscreteScheduler", "EulerDiscreteSchedulerOutput", "UNetSpatioTemporalConditionControlNetModel",
           "UNetSpatioTemporalConditionOutput"]
