import torch, sys, os
sys.path.insert(0,'/root/repo')
from oracle import vae as OV, init as OI, quant as OQ
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
torch.manual_seed(0)
for name,cfg,hw,nf in (("tiny",OV.tiny_vae_config(),(8,8),6),("svd",OV.svd_vae_config(),(8,8),3)):
    o=OI.seeded_init_(OV.AutoencoderKLTemporalDecoder(**cfg),seed=51).eval()
    with torch.no_grad():
        for p in o.parameters(): p.copy_(p.half().float())
        g=torch.Generator().manual_seed(7)
        z=(torch.randn(nf,4,*hw,generator=g)*1.2).half().float()
        ref=o.decode(z,num_frames=nf).sample
        for m in ("fp16-fused","fp16"):
            with OQ.storage(m):
                y=o.decode(z,num_frames=nf).sample
            print(name,"decode",m,rel(y,ref))
        x=(torch.rand(1,3,64,64,generator=g)*2-1).half().float()
        r=o.encode(x).latent_dist.mode()
        for m in ("fp16-fused","fp16"):
            with OQ.storage(m):
                y=o.encode(x).latent_dist.mode()
            print(name,"encode",m,rel(y,r))
