#!/usr/bin/env python3
"""dev tool: after 400 operator-form steps, is the reported post-update MSE still the per-frame form's on the same weights?"""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
D, N, maps, Nk, B = 3, 256, [8, 16, 32], 5, 8
rng = np.random.default_rng(3)
ws = []
dD = D
for dM in maps:
    ws.append((rng.uniform(-1, 1, (dM, dD, Nk, Nk)) / (dD * Nk), rng.uniform(-.1, .1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)) / (dM * Nk), rng.uniform(-.1, .1, dD)))
    dD = dM
frames = [ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N)))) for _ in range(4)]
def mk(w):
    net = aefft.Net(ctx, D, N, N, maps, Nk, 2, batch=B)
    for l, x in enumerate(w): net.set_pair(l, *x)
    return net
recon = ctx.empty(B, D, N, N); mse = ctx.empty(len(maps))
ctx.set_flags()
net = mk(ws)
for steps in (0, 50, 400):
    while getattr(net, "_n", 0) < steps:
        net.step_grad(frames[getattr(net, "_n", 0) % 4], recon); net.step_apply(0.02, 0, 0, 1.0, mse); net._n = getattr(net, "_n", 0) + 1
    ctx.sync()
    w = [net.get_pair(l) for l in range(len(maps))]
    out = {}
    for name, fl in (("operator", []), ("per-frame", ["NOOPFORM"])):
        ctx.set_flags(*fl)
        n2 = mk(w)
        n2.step_grad(frames[0], recon); n2.step_apply(0.0, 0, 0, 1.0, mse); ctx.sync()
        out[name] = mse.cpu().numpy().copy(); n2.close()
    ctx.set_flags()
    print(f"after {steps} steps: operator {out['operator']} per-frame {out['per-frame']} rel {np.abs(out['operator'] - out['per-frame']) / out['per-frame']}")
