import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
aefft = importlib.import_module("autoencoder-fft_amd"); import np_ref as R
ctx = aefft.Context(0)
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/fft_path.npz"))
tag, D, N, maps, Nk, s = "B", 3, 32, [4, 6], 3, 2
L = 2
net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=1)
for l in range(L):
    net.set_pair(l, g[f"{tag}_c{l}"], g[f"{tag}_b{l}"], g[f"{tag}_c{2*L-1-l}"], g[f"{tag}_b{2*L-1-l}"])
x = g[f"{tag}_x"]
recon = ctx.empty(1, D, N, N)
net.forward(ctx.dev(x[None]), recon)
for l in range(9):
    ref = g[f"{tag}_layer{l}"]; got = net.get_layer(l).cpu().numpy()[0]
    print(l, ref.shape, np.abs(got-ref).max()/np.abs(ref).max())
