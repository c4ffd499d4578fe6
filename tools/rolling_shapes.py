"""Rolling kernel across feature counts / bias (ms per launch, ns per row) -- run on the GPU box."""
import sys

import torch

sys.path.insert(0,'.')
import polars_ds_extension_amd as pds
dev=torch.device('cuda',0); g=torch.Generator(device=dev); g.manual_seed(1)
ctx=pds.Context(0); ctx.set_stream(torch.cuda.current_stream())
for n,p,bias in ((100_000_000,8,False),(100_000_000,6,False),(50_000_000,4,True),(30_000_000,10,False),(30_000_000,12,False)):
    xs=[torch.rand(n,dtype=torch.float64,device=dev,generator=g) for _ in range(p)]
    y=sum(xs[j]*(0.1*(j+1)) for j in range(p))+1e-3*torch.randn(n,dtype=torch.float64,device=dev,generator=g)
    pds.rolling_lin_reg(*xs,target=y,window_size=256,add_bias=bias,ctx=ctx)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    for _ in range(3): pds.rolling_lin_reg(*xs,target=y,window_size=256,add_bias=bias,ctx=ctx)
    ctx.set_timing(False); t=ctx.get_timing(reset=True)['rolling']
    print(n,p,bias,'ms',round(t[0]/t[1],3),'ns/row',round(t[0]/t[1]*1e6/n,3))
    del xs,y; torch.cuda.empty_cache()
