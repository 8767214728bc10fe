import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests import parity as P
for hw in ((16, 16), (40, 72)):
    d = P.net_ladder("cuda:0", latent_hw=hw, modes=("fp32", "fp16-fused"))
    print(os.environ.get("PT_WIDE_KINDS", "default"), hw, "unet hip|fp32 %.3e fused|fp32 %.3e   cn hip|fp32 %.3e fused|fp32 %.3e" % (
        d["unet"]["hip|fp32"], d["unet"]["fp16-fused|fp32"], d["controlnet_mid"]["hip|fp32"], d["controlnet_mid"]["fp16-fused|fp32"]), flush=True)
