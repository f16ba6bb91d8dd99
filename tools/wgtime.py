#!/usr/bin/env python3
"""dev tool (GPU box): per-workgroup start / end times inside the grouped weight-side launches of the cfg3-P2 step (an experiment build with
-DAEFFT_X_WGTIME=1: tools/mkx.sh WGT "-DAEFFT_X_WGTIME=1" opform_kernels pruned_kernels weight_kernels; AEFFT_LIB points at it).
Prints, per kernel, the launch's span and -- in 16 buckets of the workgroup index -- when the bucket's workgroups start, how long they run and
when the last of them ends: the long pole of a heterogeneous launch shows as a bucket that ends last."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import bench
aefft = importlib.import_module("autoencoder-fft_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "p2"
N, maps, s, B, sym, maxdiff = {"p2": (512, [8, 16, 32, 64], 2, 32, 0, 0), "cfg5": (1024, [8, 16, 32, 64, 128], 2, 32, 1, 1), "cfg2": (256, [8, 16, 32], 2, 1, 0, 0)}[cfg]
torch.cuda.set_device(0)
ctx = aefft.Context(0, use_torch_stream=False)
L = aefft.lib()
NS, WMAX = 5, 8192
buf = torch.zeros(NS * WMAX * 10, dtype=torch.int64, device="cuda:0")
for f in ("aefft_debug_wgtime_opform", "aefft_debug_wgtime_pruned", "aefft_debug_wgtime_weight"):
    fn = getattr(L, f); fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
net = aefft.Net(ctx, 3, N, N, maps, 5, s, batch=B); bench.init_weights(net, np)
frames = bench.synth_frames(torch, B, 3, N, "cuda:0", 0); recon = torch.empty_like(frames)
for _ in range(30): net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, None)
ctx.sync()
for f in ("aefft_debug_wgtime_opform", "aefft_debug_wgtime_pruned", "aefft_debug_wgtime_weight"):
    assert getattr(L, f)(C.c_void_p(buf.data_ptr())) == 0
for _ in range(5): net.step_grad(frames, recon); net.step_apply(0.2, maxdiff, sym, 1.0, None)
ctx.sync()
raw = buf.cpu().numpy().astype(np.float64) / 100.0                                  # microseconds (100 MHz)
t = raw[:NS * WMAX * 2].reshape(NS, WMAX, 2)
st = raw[NS * WMAX * 2:].reshape(NS, WMAX, 8)                                        # stage stamps (0 = not taken)
for k, name in enumerate(("msgrad", "kgrad", "wgrad_taps", "kspec", "tail")):
    a = t[k]; live = a[:, 1] > 0; n = int(live.sum())
    if not n: print(name, "no data"); continue
    a = a[:n]; t0 = a[:, 0].min(); span = a[:, 1].max() - t0
    print(f"{name}: {n} workgroups recorded (of the launch's first {WMAX}), span {span:.1f} us; first start -> bucket [start median, duration median / max, last end]")
    nb = 16
    for b in range(nb):
        sl = a[b * n // nb:(b + 1) * n // nb]
        if not len(sl): continue
        ss = st[k][b * n // nb:(b + 1) * n // nb]
        stages = " ".join(f"{np.median((ss[:, j] - sl[:, 0])[ss[:, j] > 0]):5.1f}" if (ss[:, j] > 0).any() else "    -" for j in range(8))
        print(f"   wg {b*n//nb:5d}..{(b+1)*n//nb-1:5d}: start {np.median(sl[:,0])-t0:6.1f}  dur {np.median(sl[:,1]-sl[:,0]):5.1f} / {(sl[:,1]-sl[:,0]).max():5.1f}  end {sl[:,1].max()-t0:6.1f}   stage stamps after start: {stages}")
