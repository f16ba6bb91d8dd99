#!/usr/bin/env python3
"""dev tool (GPU box): the four transform kernels of the cfg3 step in isolation -- r2c_pool (row pass + column pass with the crop) and
unpool_c2r (column pass + row pass with the zero-pad) on B*D planes, HIP-event brackets per kernel (aefft_prof_*).
   tools/fftbench.py [N=512] [planes=96] [scale=2] [reps=20]"""
import importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
aefft = importlib.import_module("autoencoder-fft_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 96
s = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.cuda.set_device(0)
ctx = aefft.Context(0, use_torch_stream=False)
x = torch.floor(torch.rand(planes, N, N, device="cuda:0") * 256)
X = ctx.r2c_pool(x, s)
y = ctx.unpool_c2r(X, N // s, -s, 1.0 / (N * N))
for _ in range(5):
    ctx.r2c_pool(x, s); ctx.unpool_c2r(X, N // s, -s, 1.0 / (N * N))
ctx.sync()
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(reps):
    ctx.r2c_pool(x, s); ctx.unpool_c2r(X, N // s, -s, 1.0 / (N * N))
pr = ctx.prof_read(); ctx.prof_enable(False)
print(f"lib {os.path.basename(aefft.LIB_PATH)}: N={N} planes={planes} scale={s}")
for k in ("r2c_rows", "r2c_cols", "c2r_cols", "c2r_rows"):
    v = pr[k]
    print(f"   {k:10s} {v['ms']/v['launches']*1e3:7.1f} us  {v['bytes']/v['launches']/1e6:7.1f} MB  {v['bytes']/v['ms']/1e6:6.0f} GB/s")
# the reconstruction's shape in the cfg3-P2 step: a compact 32 x 17 spectrum per plane zero-padded to 512 x 512 (the row pass is a write stream)
Xc = ctx.r2c_pool(x, 16)
for _ in range(3): ctx.unpool_c2r(Xc, N // 16, -16, 1.0 / (N * N))
ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
for _ in range(reps): ctx.unpool_c2r(Xc, N // 16, -16, 1.0 / (N * N))
pr = ctx.prof_read(); ctx.prof_enable(False)
for k in ("c2r_cols", "c2r_rows"):
    v = pr[k]
    if not v["launches"]: continue
    print(f"   recon {k:10s} {v['ms']/v['launches']*1e3:7.1f} us  {v['bytes']/v['launches']/1e6:7.1f} MB  {v['bytes']/v['ms']/1e6:6.0f} GB/s")
# correctness guard of an experiment build: round trip of a band-limited plane
xb = ctx.unpool_c2r(X, N // s, -s, 1.0 / (N * N))
X2 = ctx.r2c_pool(xb, s)
print("   round-trip rel err", float((X2 - X).abs().max() / X.abs().max()))
