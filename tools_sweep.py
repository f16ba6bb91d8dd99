#!/usr/bin/env python3
"""dev tool: run the cfg3 step under a list of AEFFT_MTILE settings in ONE process (for a rocprofv3 kernel trace);
tools_sweep_parse.py then picks the best tile per contraction of the step."""
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
aefft = importlib.import_module("autoencoder-fft_amd")
variant = os.environ.get("VARIANT", "p2")
cfgs = [None] + [f"{v},{r},{c},{k}" for v in (2, 1) for (r, c) in ((1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4), (4, 4)) for k in (1, 4) if v * r * c <= 16]
if os.environ.get("CFGS"): cfgs = [None] + os.environ["CFGS"].split(";")
ctx = aefft.Context(0)
D, N, maps, Nk, B = 3, 512, [8, 16, 32, 64], 5, 32
s = 2 if variant == "p2" else 1
net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
rng = np.random.default_rng(0)
dD = D
for l, dM in enumerate(maps):
    net.set_pair(l, rng.uniform(-1, 1, (dM, dD, Nk, Nk)) / (dD * Nk), rng.uniform(-.1, .1, dM), rng.uniform(-1, 1, (dD, dM, Nk, Nk)) / (dM * Nk), rng.uniform(-.1, .1, dD))
    dD = dM
frames = ctx.dev(np.floor(rng.uniform(0, 256, (B, D, N, N))))
recon = ctx.empty(B, D, N, N)
mse = ctx.empty(len(maps))
order = []
for cfg in cfgs:
    if cfg is None: os.environ.pop("AEFFT_MTILE", None); os.environ.pop("AEFFT_ABL", None)
    elif "/" in cfg: os.environ["AEFFT_MTILE"], os.environ["AEFFT_ABL"] = cfg.split("/")
    else: os.environ["AEFFT_MTILE"] = cfg; os.environ.pop("AEFFT_ABL", None)
    for _ in range(3):
        net.step_grad(frames, recon); net.step_apply(0.2, 0, 0, 1.0, mse)
        order.append(cfg or "auto")
    ctx.sync()
json.dump(order, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "sweep_order.json"), "w"))
