#!/usr/bin/env python3
"""dev tool (GPU box): where a SHORT timed region (the driver's --steps 20 --warmup 5) loses time against the steady state:
per-step GPU timestamps (events on the library's stream) and the host-side wall clock around the same 20 steps."""
import importlib, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
aefft = importlib.import_module("autoencoder-fft_amd")
dp = importlib.import_module("autoencoder-fft_amd.dp")
torch.cuda.set_device(0)
ctx = aefft.Context(0, use_torch_stream=False)
N, D, maps, B = 512, 3, [8, 16, 32, 64], 32
net = aefft.Net(ctx, D, N, N, maps, 5, 2, batch=B); bench.init_weights(net, np)
frames = bench.synth_frames(torch, B, D, N, "cuda:0", 0); recon = torch.empty_like(frames)
mse = torch.zeros(4, device="cuda:0")
step = dp.DataParallelStep(net)
st = ctx.torch_stream()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(3):
    for _ in range(5): step(frames, recon, 0.2, 0, 0, mse)
    ctx.sync(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record(st)
    for i in range(K):
        step(frames, recon, 0.2, 0, 0, mse); ev[i + 1].record(st); host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t0
    ctx.sync(); t_sync1 = time.perf_counter() - t0
    torch.cuda.synchronize(); t_wall = time.perf_counter() - t0
    gpu = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print(f"rep {rep}: wall {t_wall*1e6:.0f} us ({t_wall/K*1e6:.1f}/step), host enqueue done at {t_enq*1e6:.0f} us, ctx.sync returned at {t_sync1*1e6:.0f} us; "
          f"GPU first->last event {ev[0].elapsed_time(ev[K])*1e3:.0f} us; per-step GPU us: " + " ".join(f"{g:.0f}" for g in gpu))
    print("   host time after each step's enqueue (us): " + " ".join(f"{h*1e6:.0f}" for h in host))
