"""A/B on one box: the training step on one stream, with the weight gradients on a second stream, and with the spatial-loss pass on a
third (both default)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from posetraj_amd import ControlNetSDVModel, UNetSpatioTemporalConditionControlNetModel
from posetraj_amd.training import ControlNetTrainer
dev = torch.device("cuda:0")
extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("EXTRA_STREAMS", "0")))]      # occupy hardware queues first (HIP maps streams onto 4)
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=1, device=dev, keep_source=True)
cn = ControlNetSDVModel(**bench.SVD).init_random_(seed=2, device=dev, keep_source=True)
g = torch.Generator().manual_seed(9)
lat = torch.randn(1, 14, 4, 40, 72, generator=g) * 0.9; emb = torch.randn(1, 1, 1024, generator=g); maps = torch.rand(1, 14, 3, 320, 576, generator=g) * 2 - 1; mv = torch.tensor([127.0])
trs = {ws: ControlNetTrainer(dict(cn.config), cn.state_dict(), unet, learning_rate=1e-5, conditioning_dropout_prob=0.1, wgrad_stream=ws[0], spatial_stream=ws[1])
       for ws in ((False, False), (True, False), (True, True))}
for rnd in range(3):
    for ws, tr in trs.items():
        for _ in range(2): tr.step(lat, emb, mv, maps, generator=g)
        torch.cuda.synchronize(); tt = []
        for _ in range(6):
            t0 = time.perf_counter(); o = tr.step(lat, emb, mv, maps, generator=g); torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
        print(f"round {rnd} (wgrad_stream, spatial_stream)={ws!s:14}: median {sorted(tt)[3] * 1e3:7.1f} ms  min {min(tt) * 1e3:7.1f} ms  loss {o['loss']:.4f}", flush=True)
