#!/usr/bin/env python3
"""dev tool (GPU box, for rocprofv3): a few training steps of one BASELINE config.
   tools/cfgstep.py cfg5|cfg2|p2|p1 [steps] [flags]"""
import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
aefft = importlib.import_module("autoencoder-fft_amd")
CFG = {"cfg5": (1024, [8, 16, 32, 64, 128], 2, 32, 1, 1), "cfg2": (256, [8, 16, 32], 2, 1, 0, 0), "p2": (512, [8, 16, 32, 64], 2, 32, 0, 0),
       "p1": (512, [8, 16, 32, 64], 1, 32, 0, 0)}
N, maps, s, B, sym, maxdiff = CFG[sys.argv[1]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.cuda.set_device(0)
ctx = aefft.Context(0, use_torch_stream=False)
if len(sys.argv) > 3: ctx.set_flags(*sys.argv[3].split(","))
net = aefft.Net(ctx, 3, N, N, maps, 5, s, batch=B); bench.init_weights(net, np)
frames = bench.synth_frames(torch, B, 3, N, "cuda:0", 0); recon = torch.empty_like(frames)
mse = torch.zeros(len(maps), device="cuda:0")
for _ in range(2): net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, mse)
ctx.sync(); t0 = time.perf_counter()
for _ in range(steps): net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, mse)
ctx.sync(); dt = (time.perf_counter() - t0) / steps
print(f"{sys.argv[1]}: {dt*1e3:.3f} ms/step, {B/dt:.0f} frames/s, form {net.step_form()}, mse {mse.cpu().numpy()}")
if os.environ.get("PROF"):
    ctx.prof_enable(True)
    net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, mse); ctx.sync(); ctx.prof_reset()
    for _ in range(3): net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, mse)
    pr = ctx.prof_read(); ctx.prof_enable(False)
    for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]):
        if v["launches"]: print(f"   {k:14s} {v['ms']/3*1e3:8.1f} us/step  {v['launches']/3:4.1f} launches  {v['bytes']/3/1e6:9.1f} MB  {v['bytes']/max(v['ms'],1e-9)/1e6:8.0f} GB/s")
