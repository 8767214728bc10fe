"""Where the HOST time of one training step goes (cProfile over ControlNetTrainer.step, full-size networks)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from posetraj_amd import ControlNetSDVModel, UNetSpatioTemporalConditionControlNetModel
from posetraj_amd.training import ControlNetTrainer
dev = torch.device("cuda:0")
unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=1, device=dev, keep_source=True)
cn = ControlNetSDVModel(**bench.SVD).init_random_(seed=2, device=dev, keep_source=True)
g = torch.Generator().manual_seed(9)
lat = torch.randn(1, 14, 4, 40, 72, generator=g) * 0.9; emb = torch.randn(1, 1, 1024, generator=g); maps = torch.rand(1, 14, 3, 320, 576, generator=g) * 2 - 1; mv = torch.tensor([127.0])
tr = ControlNetTrainer(dict(cn.config), cn.state_dict(), unet, learning_rate=1e-5, conditioning_dropout_prob=0.1)
for _ in range(3): tr.step(lat, emb, mv, maps, generator=g)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.step(lat, emb, mv, maps, generator=g)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
