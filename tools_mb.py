import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
def bench(fn, n=30):
    for _ in range(3): fn()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(n): fn()
    p = ctx.prof_read()["contract"]; ctx.prof_enable(False)
    return p["ms"]/p["launches"]*1e3
import ast
shapes = ast.literal_eval(os.environ.get('SHAPES', '[(32,64,32,32),(16,32,64,32),(8,16,128,32),(3,8,256,32),(64,32,32,32),(32,64,32,8),(32,64,32,128)]'))
for (dD,dM,N,B) in shapes:
    X=torch.randn(B,dD,N,N//2+1,dtype=torch.complex64,device='cuda')
    Cs=torch.randn(dM,dD,N,N//2+1,dtype=torch.complex64,device='cuda')
    b=torch.randn(dM,device='cuda')
    t=bench(lambda: ctx.conv(X,Cs,b,N))
    flops=8*dM*dD*B*N*(N//2+1)
    print(os.environ.get("AEFFT_CONTRACT","auto"), (dD,dM,N,B), "%.1f us  %.2f TFLOP/s" % (t, flops/t/1e6))
