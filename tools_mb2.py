import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
dD,dM,N,B = [int(v) for v in os.environ.get("SHAPE","32,64,32,32").split(",")]
X=torch.randn(B,dD,N,N//2+1,dtype=torch.complex64,device='cuda')
Cs=torch.randn(dM,dD,N,N//2+1,dtype=torch.complex64,device='cuda')
b=torch.randn(dM,device='cuda')
for _ in range(10): ctx.conv(X,Cs,b,N)
torch.cuda.synchronize()
