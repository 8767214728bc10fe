"""Camera-disentangle variant - drop-in for ``/root/reference/models/controlnet_sdv_cam_infer.py``: the same network
with ``cc_projection`` in the condition encoder and a ``camera_cond`` argument on ``forward`` (``:537,612``)."""
from __future__ import annotations

from .controlnet_sdv import ControlNetOutput, ControlNetSDVModel as _Base


class ControlNetSDVModel(_Base):
    def __init__(self, *args, **kwargs):
        kwargs["camera"] = True
        super().__init__(*args, **kwargs)


__all__ = ["ControlNetSDVModel", "ControlNetOutput"]
