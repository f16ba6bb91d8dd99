#!/usr/bin/env python3
"""dev tool: does the input prefetch fill an idle gap between step_grad and step_apply (where the RCCL all-reduce sits at N > 1)?"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
D, N, maps, Nk, B = 3, 512, [8, 16, 32, 64], 5, 32
net = aefft.Net(ctx, D, N, N, maps, Nk, 2, batch=B)
rng = np.random.default_rng(0)
dD = D
for l, dM in enumerate(maps):
    net.set_pair(l, rng.uniform(-1, 1, (dM, dD, Nk, Nk)) / (dD * Nk), rng.uniform(-.1, .1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)) / (dM * Nk), rng.uniform(-.1, .1, dD))
    dD = dM
frames = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
recon = ctx.empty(B, D, N, N); mse = ctx.empty(len(maps))
gap_cycles = int(float(os.environ.get("GAP_US", "50")) * 2100)          # torch.cuda._sleep spins for ~cycles
ts = ctx.torch_stream()
for ready in (False, True):
    net.set_input_ready(ready)
    def step():
        net.step_grad(frames, recon)
        with torch.cuda.stream(ts):
            torch.cuda._sleep(gap_cycles)
        net.step_apply(0.2, 0, 0, 1.0, mse)
    for _ in range(10): step()
    ctx.sync(); torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K): step()
    ctx.sync(); torch.cuda.synchronize()
    print(f"input_ready={ready}: {1e6*(time.perf_counter()-t0)/K:.1f} us/step with a {os.environ.get('GAP_US','50')} us idle gap")
