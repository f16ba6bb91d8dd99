#!/usr/bin/env python3
"""dev tool: a long run of the headline step with the MSE sums deferred (step_apply without an mse output) against the same run asking
for the MSE every step: weights bit-identical at the end, MSE read every 500 steps equal, no hang over 20 000 steps."""
import importlib, sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0, use_torch_stream=False)
N, D, maps, B = 512, 3, [8, 16, 32, 64], 32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
frames = [bench.synth_frames(torch, B, D, N, "cuda:0", 32 * i) for i in range(3)]
recon = torch.empty_like(frames[0])
res = []
for ask in (False, True):
    net = aefft.Net(ctx, D, N, N, maps, 5, 2, batch=B); bench.init_weights(net, np, rmax=0.2)
    mse = torch.zeros(4, device="cuda:0"); hist = []
    t0 = time.perf_counter()
    n = steps if not ask else min(steps, 2000)
    for it in range(n):
        net.step_grad(frames[it % 3], recon); net.step_apply(0.02, 0, 0, 1.0, mse if ask else None)
        if it % 500 == 499:
            if not ask: net.last_mse(mse)
            ctx.sync(); hist.append(mse.cpu().numpy().copy())
    ctx.sync(); dt = time.perf_counter() - t0
    w = [net.get_pair(l) for l in range(4)] if n == min(steps, 2000) or ask else None
    print(f"ask={ask}: {n} steps in {dt:.2f} s = {dt/n*1e6:.1f} us/step, last mse {hist[-1]}, finite {all(np.isfinite(h).all() for h in hist)}")
    res.append((hist, w)); net.close()
h0, h1 = res[0][0], res[1][0]
print("MSE at steps 500..2000 equal:", all(np.array_equal(a, b) for a, b in zip(h0[:4], h1[:4])))
