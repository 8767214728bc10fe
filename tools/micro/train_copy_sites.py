#!/usr/bin/env python3
"""Which lines of posetraj_amd issue torch copies / casts / fills inside one training step (13 % of the step's
device time was __amd_rocclr_copyBuffer + aten cast / fill kernels): torch's copy_, clone, contiguous, to, zero_, fill_, zeros,
zeros_like, cat are wrapped for ONE step and counted by the innermost posetraj_amd frame, with the bytes they move.
    python tools/micro/train_copy_sites.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from posetraj_amd import ControlNetSDVModel, UNetSpatioTemporalConditionControlNetModel
from posetraj_amd.training import ControlNetTrainer
dev = torch.device("cuda:0")
unet = UNetSpatioTemporalConditionControlNetModel(num_attention_heads=(5, 10, 20, 20), num_frames=14).init_random_(seed=1, device=dev, keep_source=True)
cn = ControlNetSDVModel.from_unet(unet, conditioning_embedding_out_channels=(16, 32, 96, 256))
sd = cn.state_dict()
g = torch.Generator().manual_seed(3)
for k in sd:
    if k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
        sd[k] = (torch.randn(sd[k].shape, generator=g) * 0.02).half()
tr = ControlNetTrainer(dict(cn.config), sd, unet, learning_rate=1e-5, conditioning_dropout_prob=0.1)
del cn
lat = torch.randn(1, 14, 4, 40, 72, generator=g) * 0.18215 * 5
emb = torch.randn(1, 1, 1024, generator=g)
traj = torch.rand(1, 14, 3, 320, 576, generator=g) * 2 - 1
mv = torch.tensor([127.0])
gen = torch.Generator().manual_seed(5)
for _ in range(2):
    tr.step(lat, emb, mv, traj, generator=gen)
torch.cuda.synchronize()
sites = collections.Counter(); nbytes = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "posetraj_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def wrap(obj, name, size_of):
    orig = getattr(obj, name)

    def f(*a, **k):
        r = orig(*a, **k)
        try:
            t = size_of(a, k, r)
            if t is not None and t.is_cuda:
                key = f"{name:12s} {site()}"
                sites[key] += 1; nbytes[key] += t.numel() * t.element_size()
        except Exception:
            pass
        return r
    setattr(obj, name, f)
    return orig


T = torch.Tensor
saved = [(T, n, wrap(T, n, s)) for n, s in (("copy_", lambda a, k, r: a[0]), ("clone", lambda a, k, r: r), ("zero_", lambda a, k, r: a[0]), ("fill_", lambda a, k, r: a[0]),
                                            ("contiguous", lambda a, k, r: r if r.data_ptr() != a[0].data_ptr() else None),
                                            ("to", lambda a, k, r: r if (r.data_ptr() != a[0].data_ptr()) else None),
                                            ("half", lambda a, k, r: r if r.data_ptr() != a[0].data_ptr() else None),
                                            ("float", lambda a, k, r: r if r.data_ptr() != a[0].data_ptr() else None))]
saved += [(torch, n, wrap(torch, n, lambda a, k, r: r)) for n in ("zeros", "zeros_like", "cat", "stack")]
tr.step(lat, emb, mv, traj, generator=gen)
torch.cuda.synchronize()
for o, n, f in saved:
    setattr(o, n, f)
print(f"{sum(sites.values())} torch copy / cast / fill calls on device tensors in one step, {sum(nbytes.values()) / 1e9:.2f} GB touched")
for key, c in sorted(sites.items(), key=lambda kv: -nbytes[kv[0]])[:45]:
    print(f"{c:5d} x  {nbytes[key] / 1e6:9.1f} MB   {key}")
