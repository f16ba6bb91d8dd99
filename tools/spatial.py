#!/usr/bin/env python3
"""dev tool: time the spatial-mode kernels (Conv_gpu / backprop_gpu semantics) at the reference's default shape."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
aefft = importlib.import_module("autoencoder-fft_amd")
ctx = aefft.Context(0)
if os.environ.get("FLAGS"): ctx.set_flags(*os.environ["FLAGS"].split(","))
B, dD, dM, N, Nk = int(os.environ.get("B", "32")), 3, int(os.environ.get("M", "50")), int(os.environ.get("N", "256")), int(os.environ.get("NK", "3"))
rng = np.random.default_rng(0)
x = ctx.dev(np.floor(rng.uniform(0, 256, (B, dD, N, N))))
c = ctx.dev(rng.uniform(-1, 1, (dM, dD, Nk, Nk))); b = ctx.dev(rng.uniform(-1, 1, dM))
f = ctx.dev(rng.uniform(-1, 1, (dD, dM, Nk, Nk))); p = ctx.dev(rng.uniform(-1, 1, dD))
ctx.prof_enable(True)
for it in range(3):
    ctx.prof_reset()
    h = ctx.conv_spatial(x, c, b)
    o = ctx.conv_spatial(h, f, p)
    mom = [torch.zeros_like(t) for t in (c, b, f, p)]
    grads = [torch.zeros_like(t) for t in (c, b, f, p)]
    ctx.backprop_spatial(x, o, h, c, b, f, p, mom, grads, 0.2, 0.9)
    pr = ctx.prof_read()
print({k: round(v["ms"] * 1e3, 1) for k, v in pr.items() if v["launches"]})
ctx.prof_enable(False)
def step():
    h = ctx.conv_spatial(x, c, b); o = ctx.conv_spatial(h, f, p); ctx.backprop_spatial(x, o, h, c, b, f, p, mom, grads, 0.2, 0.9)
hb, ob = ctx.empty(B, dM, N, N), ctx.empty(B, dD, N, N)
def fused():
    ctx.step_spatial(x, c, b, f, p, mom, grads, 0.2, 0.9, hin=hb, out=ob)
for name, fn in (("separate calls", step), ("fused step (aefft_step_spatial)", fused)):
    for _ in range(3): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20): fn()
    ctx.sync(); print(f"step, {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms (flags {os.environ.get('FLAGS', '')})")
flops_conv = 2.0 * B * dM * dD * Nk * Nk * N * N
print(f"conv flops {flops_conv/1e9:.2f} GF each")
