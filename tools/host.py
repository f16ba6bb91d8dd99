#!/usr/bin/env python3
"""dev tool: host enqueue time vs device time of the cfg3 training step."""
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
if os.environ.get("NORECON"): recon = None
net.set_input_ready(os.environ.get("READY", "1") == "1")
for _ in range(10):
    net.step_grad(frames, recon); net.step_apply(0.2, 0, 0, 1.0, mse)
ctx.sync()
K = int(os.environ.get("K", "300"))
t0 = time.perf_counter()
for _ in range(K):
    net.step_grad(frames, recon); net.step_apply(0.2, 0, 0, 1.0, mse)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
