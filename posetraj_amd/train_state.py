"""Host side of the training loop around ``ControlNetTrainer.step``: the learning-rate schedule and checkpoint / resume
(``/root/reference/scripts/train_svd_traj_VIPSeg_14.py``: ``get_scheduler`` ``:1109-1114``, ``lr_scheduler.step()`` ``:1424``,
``accelerator.save_state`` with the ``save_model_hook`` of ``:992-1000`` at ``:1436-1467``, ``--checkpoints_total_limit`` rotation
``:1440-1462``, ``--resume_from_checkpoint`` ``:1224-1247``).

No tensor arithmetic happens here: files, counters and a scalar multiplier.  A checkpoint directory holds

    controlnet/config.json, controlnet/diffusion_pytorch_model.safetensors    what ``model.save_pretrained`` writes from the
                                                                               save hook: the fp32 master parameters, torch layout,
                                                                               readable by ``ControlNetSDVModel.from_pretrained``
    optimizer.safetensors            ``exp_avg.<name>`` / ``exp_avg_sq.<name>``: AdamW's moments per parameter, torch layout
    trainer_state.json               AdamW's step count, the loss scale and its growth counter (accelerate's ``scaler.pt``), the
                                     skipped-step count, the micro-batch position inside an accumulation cycle (always 0: states
                                     are saved after an optimizer step, like the reference does)

(the reference's ``optimizer.bin`` / ``scaler.pt`` / ``scheduler.bin`` are pickles of torch objects; the schedule here is a pure
function of the step count and needs no file).
"""
from __future__ import annotations

import json
import math
import os
import shutil
from typing import Callable, List, Optional, Tuple

SCHEDULES = ("constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial")


def get_scheduler(name: str, num_warmup_steps: int = 0, num_training_steps: Optional[int] = None, *, num_cycles: Optional[float] = None,
                  power: float = 1.0, lr_init: float = 1.0, lr_end: float = 1e-7) -> Callable[[int], float]:
    """The multiplier ``diffusers.optimization.get_scheduler(name, ...)`` applies to the base learning rate, as a function of the
    number of optimizer steps taken so far (``:1109-1114``; the script multiplies both step counts by the number of processes
    because accelerate steps the wrapped scheduler that many times per optimizer step: per optimizer step the schedule is the
    one over the unscaled counts, which is what this takes).  accelerate does not advance the schedule on a step the GradScaler
    skipped; ``ControlNetTrainer`` evaluates this at its count of steps TAKEN.  ``polynomial`` needs the base rate (``lr_init``)."""
    W, T = int(num_warmup_steps), num_training_steps
    if name not in SCHEDULES:
        raise ValueError(f"unknown lr scheduler {name!r}; one of {SCHEDULES}")
    if name not in ("constant", "constant_with_warmup") and T is None:
        raise ValueError(f"lr scheduler {name!r} needs num_training_steps")

    def warm(s):
        return s / max(1, W)

    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: warm(s) if s < W else 1.0
    if name == "linear":
        return lambda s: warm(s) if s < W else max(0.0, (T - s) / max(1, T - W))
    if name == "cosine":
        cyc = 0.5 if num_cycles is None else num_cycles

        def cosine(s):
            if s < W:
                return warm(s)
            prog = (s - W) / max(1, T - W)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * cyc * 2.0 * prog)))
        return cosine
    if name == "cosine_with_restarts":
        cyc = 1 if num_cycles is None else num_cycles

        def restarts(s):
            if s < W:
                return warm(s)
            prog = (s - W) / max(1, T - W)
            if prog >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((cyc * prog) % 1.0))))
        return restarts
    if not lr_init > lr_end:
        raise ValueError(f"lr_end ({lr_end}) must be smaller than the initial lr ({lr_init})")

    def polynomial(s):
        if s < W:
            return warm(s)
        if s > T:
            return lr_end / lr_init
        remaining = 1 - (s - W) / (T - W)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
    return polynomial


# ------------------------------------------------------------------------------------------------- checkpoint directories
def _step_of(name: str) -> int:
    return int(name.split("-")[1])


def list_checkpoints(output_dir: str) -> List[str]:
    """``checkpoint-<global_step>`` directories of ``output_dir``, oldest first (``:1441-1445``)."""
    if not os.path.isdir(output_dir):
        return []
    return sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint")), key=_step_of)


def rotate_checkpoints(output_dir: str, total_limit: Optional[int]) -> List[str]:
    """Before a new checkpoint is written: leave at most ``total_limit - 1`` old ones (``:1440-1462``).  Returns what it removed."""
    if total_limit is None:
        return []
    have = list_checkpoints(output_dir)
    if len(have) < total_limit:
        return []
    gone = have[:len(have) - total_limit + 1]
    for d in gone:
        shutil.rmtree(os.path.join(output_dir, d))
    return gone


def resolve_resume(output_dir: str, resume_from_checkpoint: Optional[str]) -> Optional[str]:
    """``--resume_from_checkpoint``: a path (its basename is looked up in ``output_dir``) or ``"latest"`` (``:1224-1233``).
    None when there is nothing to resume from (the reference then starts a new run)."""
    if not resume_from_checkpoint:
        return None
    if resume_from_checkpoint != "latest":
        name = os.path.basename(os.path.normpath(resume_from_checkpoint))
    else:
        have = list_checkpoints(output_dir)
        name = have[-1] if have else None
    if name is None or not os.path.isdir(os.path.join(output_dir, name)):
        return None
    return os.path.join(output_dir, name)


def resume_position(checkpoint_path: str, gradient_accumulation_steps: int, num_update_steps_per_epoch: int) -> Tuple[int, int, int]:
    """``(global_step, first_epoch, resume_step)`` of ``:1242-1247``: where the loop picks up (``resume_step`` = the number of
    dataloader batches of the first epoch to skip)."""
    global_step = _step_of(os.path.basename(os.path.normpath(checkpoint_path)))
    resume_global_step = global_step * gradient_accumulation_steps
    first_epoch = global_step // num_update_steps_per_epoch
    resume_step = resume_global_step % (num_update_steps_per_epoch * gradient_accumulation_steps)
    return global_step, first_epoch, resume_step


# ------------------------------------------------------------------------------------------------- trainer state <-> files
def save_controlnet(trainer, path: str) -> None:
    """``controlnet.save_pretrained(path)``: config + the fp32 master parameters in the reference's state-dict format."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = {k: v for k, v in dict(trainer.config).items() if not k.startswith("_")}
    cfg["_class_name"] = "ControlNetSDVModel"
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    save_file({k: v.cpu() for k, v in trainer.params.state_dict().items()}, os.path.join(path, "diffusion_pytorch_model.safetensors"))


def save_state(trainer, output_dir: str) -> None:
    """``accelerator.save_state(output_dir)`` for a ``ControlNetTrainer`` (``:1464-1466``).  Call it on the main process, after an
    optimizer step (the reference saves inside ``if accelerator.sync_gradients``)."""
    from safetensors.torch import save_file
    if trainer._micro != 0:
        raise RuntimeError("save_state inside an accumulation cycle: the accumulated gradients are not part of a checkpoint")
    os.makedirs(output_dir, exist_ok=True)
    save_controlnet(trainer, os.path.join(output_dir, "controlnet"))
    P = trainer.params
    moments = {}
    for kind, buf in (("exp_avg", P.exp_avg), ("exp_avg_sq", P.exp_avg_sq)):
        for k, v in P.export(buf).items():
            moments[f"{kind}.{k}"] = v.cpu()
    save_file(moments, os.path.join(output_dir, "optimizer.safetensors"))
    state = {"format": 1, "optimizer_steps": trainer.optimizer_steps, "skipped_steps": trainer.skipped_steps, "loss_scale": trainer.loss_scale,
             "growth_tracker": trainer._clean, "growth_interval": trainer.growth_interval, "micro_batch": trainer._micro,
             "hyperparameters": {"learning_rate": trainer.lr, "adam_beta1": trainer.betas[0], "adam_beta2": trainer.betas[1],
                                 "adam_weight_decay": trainer.weight_decay, "adam_epsilon": trainer.eps,
                                 "gradient_accumulation_steps": trainer.accumulation}}
    tmp = os.path.join(output_dir, "trainer_state.json.tmp")
    with open(tmp, "w") as f:
        json.dump(state, f, indent=2)
    os.replace(tmp, os.path.join(output_dir, "trainer_state.json"))      # written last: its presence marks a complete checkpoint


def load_state(trainer, input_dir: str) -> dict:
    """``accelerator.load_state(input_dir)`` (``:1241``; the ``load_model_hook`` of ``:1002-1020`` reads ``controlnet/``): parameters,
    AdamW moments, step count and loss-scale state.  Hyperparameters stay the constructor's (the script passes them again on the
    command line); the stored ones are returned for the caller to compare."""
    from safetensors.torch import load_file
    with open(os.path.join(input_dir, "trainer_state.json")) as f:
        state = json.load(f)
    if state.get("format") != 1:
        raise RuntimeError(f"{input_dir}: unknown trainer_state format {state.get('format')!r}")
    P = trainer.params
    weights = load_file(os.path.join(input_dir, "controlnet", "diffusion_pytorch_model.safetensors"))
    moments = load_file(os.path.join(input_dir, "optimizer.safetensors"))
    P.load(P.flat, weights)
    P.load(P.exp_avg, {k[len("exp_avg."):]: v for k, v in moments.items() if k.startswith("exp_avg.")})
    P.load(P.exp_avg_sq, {k[len("exp_avg_sq."):]: v for k, v in moments.items() if k.startswith("exp_avg_sq.")})
    P.version += 1                                                         # layers re-pack their fp16 operands from the new master
    P.zero_grad()
    trainer.optimizer_steps, trainer.skipped_steps = int(state["optimizer_steps"]), int(state["skipped_steps"])
    trainer.loss_scale, trainer._clean = float(state["loss_scale"]), int(state["growth_tracker"])
    trainer._micro, trainer._accum_scale = 0, None
    return state
