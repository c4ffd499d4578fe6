import sys; from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, polars_ds_extension_amd as pds
n,p,w=20_000_000,8,256
dev=torch.device("cuda",0); gen=torch.Generator(device=dev); gen.manual_seed(3)
xs=[torch.rand(n,dtype=torch.float64,device=dev,generator=gen) for _ in range(p)]
y=sum(xs[j]*(0.1*(j+1)) for j in range(p))+1e-3*torch.randn(n,dtype=torch.float64,device=dev,generator=gen)
for _ in range(3): pds.rolling_lin_reg(*xs,target=y,window_size=w)
torch.cuda.synchronize()
