#!/usr/bin/env python3
"""dev tool: 400 training steps in operator form and in the per-frame form from the same start; prints the MSE histories side by side."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
D, N, maps, Nk, B = 3, 256, [8, 16, 32], 5, 8
def run(flags, steps, del0):
    ctx.set_flags(*flags)
    net = aefft.Net(ctx, D, N, N, maps, Nk, 2, batch=B)
    rng = np.random.default_rng(3)
    dD = D
    for l, dM in enumerate(maps):
        net.set_pair(l, rng.uniform(-1, 1, (dM, dD, Nk, Nk)) / (dD * Nk), rng.uniform(-.1, .1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)) / (dM * Nk), rng.uniform(-.1, .1, dD))
        dD = dM
    frames = [ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N)))) for _ in range(4)]
    recon = ctx.empty(B, D, N, N); mse = ctx.empty(len(maps))
    hist = []
    for it in range(steps):
        net.step_grad(frames[it % 4], recon); net.step_apply(del0, 0, 0, 1.0, mse)
        if it % 50 == 49 or it == steps - 1:
            ctx.sync(); hist.append(mse.cpu().numpy().copy())
    ctx.sync()
    w = [net.get_pair(l) for l in range(len(maps))]
    r = recon.cpu().numpy().copy()
    net.close()
    return hist, w, r
for del0 in (0.02,):
    h1, w1, r1 = run([], 400, del0)
    h2, w2, r2 = run(["NOOPFORM"], 400, del0)
    for a, b in zip(h1, h2):
        print("mse op", a, "per-frame", b, "rel", np.abs(a - b) / np.maximum(np.abs(b), 1e-30))
    print("finite", all(np.isfinite(x).all() for x in h1), "recon rel diff", np.abs(r1 - r2).max() / np.abs(r2).max())
    for l in range(len(maps)):
        print("pair", l, "max |dw|", max(np.abs(x - y).max() for x, y in zip(w1[l], w2[l])))
