#!/usr/bin/env python3
"""bench.py -- frames/s forward+backward of the FFT-mode autoencoder training path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        (N > 1: either launched by torch.distributed.run, one rank per GPU, or -- run plainly -- it starts its N ranks itself)

Workload (BASELINE.json configs[2], SURVEY 8d "config 3"): 512x512 RGB frames, 4 encoder/decoder
pairs (3->8->16->32->64 maps, 5x5 kernels), FFT mode, 32 synthetic frames per GPU resident in HBM.
Default variant P2 = the reference's own pooling (New_Layer_Param.txt: scale 2 per layer);
`--variant p1` is the no-pooling, honestly HBM-bound variant.

One step = 1x autoenc_fft over the batch + for EVERY pair one backprop_fft loop-body iteration
(gradient -> C2R -> shrink -> [all-reduce of the packed kernel-support gradients when N>1] ->
update -> pad -> R2C -> re-forward of the pair -> MSE), with these outputs produced every step: the
reconstruction of every frame, the batch-mean gradients, the updated weights and kernel spectra, the
post-update MSE of every pair.  What is NOT formed per frame: the network is linear (identity activation),
so the step runs in OPERATOR FORM -- the per-bin operators of the layers on 4 basis
frames, the batch's second moments, and small per-bin matrix products in place of the per-frame Hadamard
products; intermediate feature maps are materialised only on request (aefft_net_get_layer[s]).  The same
sums as the reference, batch contracted first; `--flags NOOPFORM` runs the per-frame form, and the parity
tests run both against the oracle.

Prints ONE JSON line (rank 0): metric/value per the driver contract, plus
  "roofline":     the dominant kernel's algorithmic bytes / its HIP-event-timed duration vs 8 TB/s
  "cpu_baseline": the reference's CPU path (oracle/_ref when built, else the C port) timed on this
                  host on a bounded sample (rank 0, N=1 only)
  "variants":     short extra runs (rank 0, N=1 only): the same workload in the per-frame form, cfg3-P1, and a spatial-mode
                  Conv+Conv+backprop step
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# kernel id (aefft_prof_name) -> kernel family in the rocprofv3 --pmc summaries under profiles/
KERNEL_NAMES = {"contract": "contract_mfma_kernel<*> (per-bin channel contraction)", "r2c_rows": "r2c_rows_kernel<N> (input transform, row pass)",
                "c2r_rows": "c2r_rows_kernel<N> (reconstruction, row pass)", "r2c_cols": "fwd_cols_kernel<N,CW> (input transform, column pass + crop)",
                "c2r_cols": "inv_cols_kernel<N,CW> (reconstruction, column pass, operator expansion on load)",
                "kgrad": "kgrad_group_kernel<9,9> (pruned inverse transform of S)", "opmse": "tail_kernel (post-update MSE through G' + the next step's operator chain)",
                "chain": "chain_kernel (network on the basis frames)", "kspec": "kspec_group_kernel (G' / compact spectra / bin-major record from the taps)",
                "sgrad": "msgrad_kernel (batch moments + S of every pair)", "weight_taps": "wgrad_taps_kernel (g_c, g_f from Q in coordinate space)",
                "update": "update_group_kernel (clipped-momentum update, tied / multiobjective terms)", "gradient_diff": "gradient_diff kernels (multiobjective term)"}
PMC_FAMILY = {"contract": ["contract_mfma_kernel", "contract_fast_kernel", "contract_kernel", "contract_group_kernel"], "r2c_rows": ["r2c_rows_kernel"],
              "r2c_cols": ["fwd_cols_kernel"], "c2r_cols": ["inv_cols_kernel"], "c2r_rows": ["c2r_rows_kernel"],
              "kgrad": ["kgrad_kernel", "kgrad_group_kernel"], "kspec": ["kspec_kernel", "kspec_group_kernel"], "diff_mse": ["diff_mse_kernel"],
              "opmse": ["opmse_kernel", "tail_kernel"], "chain": ["chain_kernel"], "sgrad": ["sgrad_kernel", "msgrad_kernel"], "moment": ["moment_kernel"], "weight_taps": ["wgrad_taps_kernel"]}


def pmc_traffic(variant, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH doubled per the gfx950 note; tools/pmc.py).  None if absent."""
    path = os.path.join(ROOT, "profiles", f"r04_pmc_traffic_cfg3{variant}.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    n = b = 0
    for fam in PMC_FAMILY.get(kernel, []):
        if fam in d:
            n += d[fam]["launches"]; b += d[fam]["hbm_bytes_per_launch"] * d[fam]["launches"]
    return b / n if n else None


def finite_json(o):
    """JSON has no NaN / Infinity: a non-finite number is printed as a string (and flagged by the mse_finite fields)."""
    if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
        return str(o)
    if isinstance(o, dict):
        return {k: finite_json(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [finite_json(v) for v in o]
    return o


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preheat", type=int, default=0, help="EXTRA untimed steps in front of the W warm-up steps (reported as preheat_steps; 0 = the contract's protocol)")
    ap.add_argument("--steady-steps", type=int, default=200, help="steps of the second timed region behind the headline one (reported as steady_state; 0 = skip)")
    ap.add_argument("--force-rccl-step", action="store_true", help="N = 1 with --rccl: run the timed loop through libaefft_dp.so as N > 1 does (rehearsal of the multi-GPU host path on one GPU)")
    ap.add_argument("--torch-dist-loop", action="store_true", help="N > 1: all-reduce through torch.distributed (dp.DataParallelStep) instead of libaefft_dp.so")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU (weak scaling)")
    ap.add_argument("--variant", choices=["p2", "p1"], default="p2")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch", action="store_true", help="aefft_net_set_input_ready: run the input R2C on a side stream ahead of the context stream")
    ap.add_argument("--cu-split", type=int, default=0, help="aefft_ctx_partition: compute units of the side streams (0 = no partition)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dp-probe", action="store_true", help="N = 1: skip the data_parallel block (a one-rank RCCL communicator, 20 untimed steps)")
    ap.add_argument("--no-variants", action="store_true", help="skip the short extra runs (cfg3-P1, spatial mode) reported under \"variants\"")
    ap.add_argument("--flags", default="", help="comma-separated development switches (include/aefft.h AEFFT_F_*), e.g. NOOVERLAP,NOOPFORM")
    ap.add_argument("--rccl", action="store_true", help="initialise the nccl (RCCL) process group even at N = 1")
    ap.add_argument("--torch-stream", action="store_true", help="enqueue on torch's current (legacy default) stream instead of a private stream")
    return ap.parse_args()


def synth_frames(torch, B, D, N, device, first_index):
    """per-pixel floor(U[0,256)) + a smooth low-frequency component, seed = 1000 + frame index."""
    out = torch.empty(B, D, N, N, dtype=torch.float32, device=device)
    i = torch.arange(N, device=device, dtype=torch.float32)[:, None] / N
    j = torch.arange(N, device=device, dtype=torch.float32)[None, :] / N
    for b in range(B):
        g = torch.Generator(device=device); g.manual_seed(1000 + first_index + b)
        noise = torch.floor(torch.rand(D, N, N, generator=g, device=device) * 256)
        for d in range(D):
            smooth = 64 * (1 + torch.sin(2 * torch.pi * (i * (d + 1) + 0.3))) * (1 + torch.cos(2 * torch.pi * j * 2)) / 2
            out[b, d] = torch.floor(0.5 * noise[d] + smooth)
    return out


def init_weights(net, np, rmax=3.0):
    """U(-rmax, rmax), rmax = 3 (New_Layer_Param.txt:5), fixed seeds -> identical on every rank."""
    for l, g in enumerate(net.dims):
        rng = np.random.default_rng(4242 + l)
        c = rng.uniform(-rmax, rmax, (g["dM"], g["dD"], g["Nk"], g["Nl"]))
        f = rng.uniform(-rmax, rmax, (g["dD"], g["dM"], g["Nk"], g["Nl"]))
        net.set_pair(l, c, rng.uniform(-rmax, rmax, g["dM"]), f, rng.uniform(-rmax, rmax, g["dD"]))


def cpu_baseline(N, scale, maps=(8, 16, 32, 64), D=3, Nk=5):
    """The reference's CPU path (netlib.cpp Pool/Conv/Conv/Pool/backprop; oracle/_ref when built, else the C port) on this
    host, BASELINE.md section 2: pair 1 of the workload (3->8 maps, 5x5) for ONE frame (i) single-threaded, as the reference
    runs, and (ii) one frame per core over all host cores; pairs 2-4 are extrapolated from the loop nest's iteration count
    (dM*dD^2*(Nk*Nl)^2*Nx*Ny for backprop, netlib.cpp:361-451 -- 3.6x, 7.2x, 14.5x pair 1) and labelled as such."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu
    cores = min(len(os.sched_getaffinity(0)), 16)
    ctx = mp.get_context("spawn")               # never fork a process that holds a HIP context
    os.environ["PYTHONPATH"] = os.path.join(ROOT, "oracle") + os.pathsep + os.environ.get("PYTHONPATH", "")   # for the workers
    job = (N, scale, D, maps[0], Nk, 1)
    with ctx.Pool(1) as pool:
        t1, kind = pool.map(cpu.time_pair_frame, [job])[0]
    with ctx.Pool(cores) as pool:
        ts = pool.map(cpu.time_pair_frame, [(N, scale, D, maps[0], Nk, 1 + i) for i in range(cores)], chunksize=1)
    wall = max(t for t, _ in ts)                # the workers run side by side; process start-up is not the reference's cost
    # iteration counts of the backprop nest per pair (dominant term) relative to pair 1
    dims, dD, n = [], D, N
    for dM in maps:
        n //= scale
        dims.append(dM * dD * dD * (Nk * Nk) ** 2 * n * n); dD = dM
    rel = [d / dims[0] for d in dims]
    full_1t = t1 * sum(rel)
    return {"value": cores / wall, "unit": "frames/s (pair 1 of 4 only)", "cores": cores, "kind": kind,
            "sample": f"{cores} frames {N}x{N}x3, one per core, pair 1 only (3->{maps[0]} maps, {Nk}x{Nk}, pool {scale}): "
                      f"Pool+Conv+Conv+Pool+backprop, {wall:.1f} s (slowest worker)",
            "single_thread": {"value": 1.0 / t1, "unit": "frames/s (pair 1 of 4 only)", "cores": 1, "seconds_per_frame": t1,
                              "sample": "1 frame, pair 1 only, single-threaded as the reference runs"},
            "extrapolated_all_pairs": {"seconds_per_frame_single_thread": full_1t, "frames_per_s_single_thread": 1.0 / full_1t,
                                       "frames_per_s_all_cores": (cores / wall) / sum(rel), "relative_cost_per_pair": rel,
                                       "method": "EXTRAPOLATED, not timed: pair-1 time x iteration-count ratio of the backprop loop nest"}}


def fft_variant(aefft, torch, np, ctx, label, N, maps, scale, B, steps, warmup=2, sym=0, maxdiff=0, flags=(), del0=0.2, u8=False):
    """A short timed run of one BASELINE config through the same step the headline uses, with its own HIP-event profile:
    frames/s, algorithmic bytes of the kernel groups as launched, whole-step fraction of the HBM peak, the dominant
    kernel's roofline object, and the first / last per-pair post-update MSE of the run (a non-finite value is FLAGGED, the
    measurement stands: with del0 = 0.2 and U(-3,3) weights the trajectory is in the reference's own unstable regime)."""
    D, Nk = 3, 5
    ctx.set_flags(*flags)
    try:
        net = aefft.Net(ctx, D, N, N, list(maps), Nk, scale, batch=B)
        form = net.step_form()
        init_weights(net, np)
        dev = f"cuda:{ctx.device}"
        frames = synth_frames(torch, B, D, N, dev, first_index=0)
        recon = torch.empty_like(frames)
        if u8:
            frames = frames.to(torch.uint8)      # (the synthetic pixels are integers 0..255: the same values, a quarter of the bytes)
        mse = torch.zeros(len(maps), dtype=torch.float32, device=dev)

        def step():
            net.step_grad(frames, recon); net.step_apply(del0, maxdiff, sym, 1.0, None)

        step(); net.last_mse(mse); ctx.sync()
        mse_first = mse.cpu().numpy().tolist()
        for _ in range(max(warmup - 1, 0)):
            step()
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        net.last_mse(mse); ctx.sync()
        mse_last = mse.cpu().numpy().tolist()
        ctx.prof_enable(True)
        step()                                  # (first launches of the profiled pass's own kernels: excluded)
        ctx.sync(); ctx.prof_reset()
        for _ in range(2):
            step()
        prof = ctx.prof_read(); ctx.prof_enable(False)
        net.close()
        del frames, recon
        torch.cuda.empty_cache()
    finally:
        ctx.set_flags()
    step_bytes = sum(v["bytes"] for v in prof.values()) / 2
    name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_s = dom["ms"] / dom["launches"] * 1e-3
    ach = dom["bytes"] / dom["launches"] / avg_s / 1e9 if avg_s else 0.0
    roof = {"bound": "hbm", "kernel": KERNEL_NAMES.get(name, name), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "avg_us": avg_s * 1e6, "launches_per_step": dom["launches"] / 2, "algo_bytes_per_launch": dom["bytes"] / dom["launches"],
            "share_of_kernel_time": dom["ms"] / sum(v["ms"] for v in prof.values()), "traffic": None}
    if name == "gradient_diff":
        # the multiobjective term is arithmetic, not traffic: per kernel pair 2 Nk*Nl FMAs (the dot product of the expanded squared distance and the
        # weighted partner sum), a reciprocal and three more operations (fft_backproplib.cu:709-753 re-associated, update_kernels.hip) against the fp32
        # vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s); c and f each
        dD, flops = 3, 0.0
        for dM in maps:
            flops += 2.0 * (dM * dD) ** 2 * (4.0 * Nk * Nk + 4.0); dD = dM
        t_s = dom["ms"] / 2 * 1e-3
        roof = {"bound": "valu", "kernel": KERNEL_NAMES.get(name, name), "achieved": flops / t_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                "frac": flops / t_s / 1e12 / 157.3, "avg_us": avg_s * 1e6, "launches_per_step": dom["launches"] / 2, "algo_flops_per_step": flops,
                "share_of_kernel_time": dom["ms"] / sum(v["ms"] for v in prof.values()), "traffic": None}
    return {"workload": label, "step_form": form, "frames_per_s": B / dt, "ms_per_step": dt * 1e3, "steps": steps, "mse_first": mse_first, "mse_last": mse_last,
            "mse_finite": bool(np.isfinite(mse_last).all()), "step_algo_GB": step_bytes / 1e9, "step_frac_of_hbm_peak": step_bytes / dt / 1e9 / HBM_PEAK_GBS,
            "roofline": roof}


def variant_spatial(aefft, torch, np, ctx, steps=20):
    """Spatial mode (a13-a15): Conv_gpu -> Conv_gpu -> backprop_gpu on 32 frames 256x256x3, 50 maps, 3x3 (the reference's default
    layer, New_Layer_Param.txt), as ONE call (aefft_step_spatial: the hidden layer is the call's own convolution, so dF and dP come out of
    the error-input region sums like dC and dB and the 50-plane hidden layer is read once); the three separate calls -- what the vector shims
    run, with a caller-supplied hidden layer -- are timed beside it.  Flops roofline: 2*dM*dD*Nk*Nl*Nx*Ny per conv, fp32 matrix-core peak
    157.3 TFLOP/s (MI355X_MICROARCH.md).  The convs are bound by their 419 MB output / input streams, not by flops."""
    B, dD, dM, N, Nk = 32, 3, 50, 256, 3
    rng = np.random.default_rng(0)
    x = ctx.dev(np.floor(rng.uniform(0, 256, (B, dD, N, N))))
    c = ctx.dev(rng.uniform(-1, 1, (dM, dD, Nk, Nk))); b = ctx.dev(rng.uniform(-1, 1, dM))
    f = ctx.dev(rng.uniform(-1, 1, (dD, dM, Nk, Nk))); p = ctx.dev(rng.uniform(-1, 1, dD))
    mom = [torch.zeros_like(t) for t in (c, b, f, p)]
    grads = [torch.zeros_like(t) for t in (c, b, f, p)]
    hbuf = ctx.empty(B, dM, N, N); obuf = ctx.empty(B, dD, N, N)

    def fused():
        ctx.step_spatial(x, c, b, f, p, mom, grads, 0.2, 0.9, hin=hbuf, out=obuf)

    def separate():
        h = ctx.conv_spatial(x, c, b)
        o = ctx.conv_spatial(h, f, p)
        ctx.backprop_spatial(x, o, h, c, b, f, p, mom, grads, 0.2, 0.9)

    def timed(step):
        for _ in range(5):
            step()
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    dt_sep = timed(separate)
    dt = timed(fused)
    conv_flops = 2.0 * B * dM * dD * Nk * Nk * N * N
    # work of the groups AS LAUNCHED: conv 3->50, conv 50->3 and the error-input correlation behind dC, dB, dF, dP (dD x dD x 25 sums per pixel)
    flops = 2 * conv_flops + 2.0 * B * dD * dD * 25 * N * N
    act, hid = 4.0 * B * dD * N * N, 4.0 * B * dM * N * N        # bytes of a 3-plane and of the 50-plane tensor
    algo_bytes = (act + hid) + (hid + act) + 2 * act              # conv, conv, error-input correlation (out, in)
    sep_bytes = algo_bytes + (hid + 2 * act)                      # ... + the dF correlation of the separate calls (hin, out, in)
    return {"workload": f"spatial mode: Conv_gpu+Conv_gpu+backprop_gpu in one call (aefft_step_spatial), {B} frames {N}x{N}x{dD}, {dM} maps, {Nk}x{Nk}", "frames_per_s": B / dt,
            "ms_per_step": dt * 1e3, "steps": steps, "algo_TFLOP_per_step": flops / 1e12, "step_algo_GB": algo_bytes / 1e9,
            "roofline": {"bound": "hbm", "achieved": algo_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo_bytes / dt / 1e9 / HBM_PEAK_GBS,
                         "note": "whole step; the binding bound: the 50-map hidden layer (419 MB) is written once and read once"},
            "mfma_roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / dt / 1e12 / 157.3},
            "separate_calls": {"ms_per_step": dt_sep * 1e3, "frames_per_s": B / dt_sep, "step_algo_GB": sep_bytes / 1e9,
                               "frac_of_hbm_peak": sep_bytes / dt_sep / 1e9 / HBM_PEAK_GBS,
                               "note": "aefft_conv_spatial x2 + aefft_backprop_spatial with a caller-supplied hidden layer (what the vector shims run): the hidden layer is read twice"}}


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a torch.distributed launcher: start the N ranks ourselves (one process per GPU) as a
    CHILD torch.distributed.run and pass its exit code on.  Nothing in this process has touched the GPU yet
    (torch.cuda.device_count() does not initialise it), and it never re-execs."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} requested but only {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    # ONE JSON line on stdout: libraries that print banners on file descriptor 1 (RCCL prints its version block at communicator
    # creation) go to stderr for the whole run; the line itself is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.rccl:
        # RCCL ("nccl" backend) also at world size 1 when asked: the all-reduce then runs through the same library path
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        assert dist.get_world_size() == world
    aefft = importlib.import_module("autoencoder-fft_amd")
    ctx = aefft.Context(local, use_torch_stream=a.torch_stream)
    if a.flags:
        ctx.set_flags(*a.flags.split(","))
    if a.cu_split:
        ctx.partition(a.cu_split)
    N, D, maps, Nk = a.size, 3, [8, 16, 32, 64], 5
    s = 2 if a.variant == "p2" else 1
    B = a.batch
    net = aefft.Net(ctx, D, N, N, maps, Nk, s, batch=B)
    init_weights(net, np)
    dev = f"cuda:{local}"
    frames = synth_frames(torch, B, D, N, dev, first_index=rank * B)      # each rank its own shard of the global batch
    recon = torch.empty_like(frames)
    torch.cuda.synchronize()                                              # inputs resident before anything is timed
    mse = torch.zeros(len(maps), dtype=torch.float32, device=dev)
    dp = importlib.import_module("autoencoder-fft_amd.dp")
    if a.prefetch or world > 1:
        # the synthetic frames are complete in HBM before the timed region: the next step's input R2C may run on a side stream.
        # Data-parallel runs wait for the gradient all-reduce between the two halves of a step; the prefetched R2C fills that gap
        # together with this step's reconstruction inverse FFT (tools/gap.py: a 40 us gap then costs +11 us per step instead of +32;
        # without a gap the mode costs +2 us, hence off at N = 1)
        net.set_input_ready(True)
    del0 = 0.2                                                            # autoencoder.cpp:87
    # The step's host side.  N = 1: step_grad -> step_apply, nothing in between.  N > 1: libaefft_dp.so (include/aefft_dp.h) enqueues
    # step_grad -> ncclAllReduce(SUM) on the library's stream -> step_apply(1/world) in ONE C call: no Python, no torch.distributed inside
    # the step (torch.distributed only carries the ncclUniqueId, the barriers around the timed region and the max over ranks).
    rstep = None
    if (world > 1 or (a.force_rccl_step and dist is not None)) and not a.torch_dist_loop:
        def bcast(idbuf):
            t = idbuf.to(dev)
            dist.broadcast(t, 0)
            return t
        rstep = dp.RcclStep(net, rank, world, bcast)
    dpstep = dp.DataParallelStep(net)                                     # (--torch-dist-loop, and N = 1 where it is step_grad -> step_apply)

    def step():
        # (no per-step MSE output: the per-pair sums of a step are then formed by one more workgroup of the next step's gradient launch --
        # in time for the all-reduce that carries them -- instead of a launch of their own; net.last_mse() delivers them where they are read)
        if rstep is not None:
            rstep(frames, recon, del0)
        else:
            dpstep(frames, recon, del0, 0, 0, None)

    def fence():
        ctx.sync()                      # the library's stream
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The contract's protocol: W untimed warm-up steps, then exactly K timed steps between fences (--preheat adds EXTRA untimed steps in
    # front and says so in the line: preheat_steps; default 0).  After idling through the set-up the chip takes ~10 ms of sustained load to
    # reach its steady clocks (tools/steptimes.py: the first ~40 steps run 4-6 % slow), so a 20-step timed region reads that ramp; the
    # SECOND timed region below (steady_state: --steady-steps more steps right behind the first) reads the step without it.
    mse_first = None
    for i in range(a.preheat + a.warmup):      # (the last a.warmup of them are the contract's W warm-up steps)
        step()
        if i == 0:
            net.last_mse(mse); ctx.sync(); mse_first = mse.cpu().numpy().tolist()      # post-update MSE of the very first step (untimed)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steady = None
    if a.steady_steps > 0:
        fence()
        t1 = time.perf_counter()
        for _ in range(a.steady_steps):
            step()
        fence()
        dts = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        steady = {"steps": a.steady_steps, "ms_per_step": dts / a.steady_steps * 1e3, "frames_per_s": world * B * a.steady_steps / dts,
                  "note": "second timed region, right behind the headline one (same fences, same step): the chip at its steady clocks"}
    net.last_mse(mse)
    ctx.sync()
    mse_host = mse.cpu().numpy().tolist()
    dp_diag = None
    if rstep is not None or (world == 1 and not a.no_dp_probe and dist is None):
        # Data-parallel diagnostics OUTSIDE the timed regions, through libaefft_dp.so: 20 more steps with events on the library's stream
        # around the gradient half, the collective and the update half, and the HOST's time per step (enqueue only, no synchronisation).
        # At N = 1 a communicator of one rank is created just for this: the all-reduce then runs through RCCL on the library's stream
        # exactly as on 8 GPUs, and the line says what the collective's launch + the host path cost before the hardware shows up.
        probe = rstep if rstep is not None else dp.RcclStep(net, 0, 1)
        probe.run(frames, recon, del0, 3)
        ph, host_us = probe.profile(frames, recon, del0, 20)
        gm = probe.flush_mse().tolist()
        agree = probe.replicas_agree()
        allph = torch.tensor(ph + [host_us], dtype=torch.float64, device=dev)[None]
        if world > 1:
            g = [torch.zeros_like(allph) for _ in range(world)]
            dist.all_gather(g, allph)
            allph = torch.cat(g)
        allph = allph.cpu().numpy()
        mm = lambda c: {"min": float(allph[:, c].min()), "max": float(allph[:, c].max())}
        dp_diag = {"steps": 20, "path": "libaefft_dp.so: step_grad -> ncclAllReduce on the library's stream -> step_apply in one C call per step",
                   "grad_half_ms": mm(0), "allreduce_ms": mm(1), "apply_half_ms": mm(2), "host_us_per_step": mm(3),
                   "host_budget_us_per_step": 100.0, "allreduce_bytes": probe.allreduce_bytes(), "replicas_agree": bool(agree),
                   "global_batch_mse_per_pair": gm, "rccl_world": world,
                   "note": "events on the library's stream; min / max over ranks; host_us = wall time the host spends enqueueing one step"}
        if rstep is None:
            probe.close()
    elif dist is not None:
        # Data-parallel diagnostics, OUTSIDE the timed region (20 more steps with events on the library's stream around the two halves and
        # the collective): per-phase milliseconds as min / max over ranks, whether the replicas still hold identical weights, and the
        # global-batch MSE (the floats that ride in the gradient all-reduce)
        dpstep.timers = []
        for _ in range(20):
            step()
        fence()
        ph = torch.tensor(dpstep.phase_ms(), dtype=torch.float64, device=dev)
        dpstep.timers = None
        allph = [torch.zeros_like(ph) for _ in range(world)]
        dist.all_gather(allph, ph)
        allph = torch.stack(allph).cpu().numpy()
        agree = bool(dpstep.replicas_agree())
        gm = dpstep.flush_mse().cpu().numpy().tolist()
        dp_diag = {"steps": 20, "grad_half_ms": {"min": float(allph[:, 0].min()), "max": float(allph[:, 0].max())},
                   "allreduce_ms": {"min": float(allph[:, 1].min()), "max": float(allph[:, 1].max())},
                   "apply_half_ms": {"min": float(allph[:, 2].min()), "max": float(allph[:, 2].max())},
                   "allreduce_bytes": int(dpstep.gbuf.numel() * 4), "replicas_agree": agree, "global_batch_mse_per_pair": gm,
                   "note": "events on the library's stream; the phases of one rank add up to its step (the reconstruction overlaps the grad half)"}
    mse_finite = bool(np.isfinite(mse_host).all())      # flagged in the JSON line, the timing stands (del0 = 0.2 on U(-3,3) weights is the reference's unstable regime)

    roof = None
    if rank == 0 and not a.no_roofline:
        # per-kernel HIP events on the stream the kernels run on, over the same step
        # (side streams off: overlapped kernels would be charged each other's time)
        ctx.prof_enable(True)
        for _ in range(2):              # (the profiled pass launches kernels the overlapped pass does not -- e.g. the update as its own launch: first launches excluded)
            net.step_grad(frames, recon); net.step_apply(del0, 0, 0, 1.0 / world, mse)
        ctx.sync(); ctx.prof_reset()
        PS = 20                          # profiled steps (an average over 3 moved by +-15 % from run to run)
        for _ in range(PS):
            net.step_grad(frames, recon); net.step_apply(del0, 0, 0, 1.0 / world, mse)
        prof = ctx.prof_read(); ctx.prof_enable(False)
        tot = sum(v["ms"] for v in prof.values())
        name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        per_launch_bytes = dom["bytes"] / dom["launches"]
        avg_s = dom["ms"] / dom["launches"] * 1e-3
        ach = per_launch_bytes / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": KERNEL_NAMES.get(name, name), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(a.variant, name) if (a.size == 512 and a.batch == 32) else None, "avg_us": avg_s * 1e6, "launches_per_step": dom["launches"] / PS,
                "algo_bytes_per_launch": per_launch_bytes, "share_of_kernel_time": dom["ms"] / tot,
                "kernels": {k: {"ms_per_step": v["ms"] / PS, "launches_per_step": v["launches"] / PS,
                                "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else 0.0}
                            for k, v in prof.items() if v["launches"]}}
        # whole-step view: algorithmic bytes of all launched kernel groups / wall time of the timed region
        step_bytes = sum(v["bytes"] for v in prof.values()) / PS
        roof["step_algo_GB"] = step_bytes / 1e9
        roof["step_frac_of_hbm_peak"] = step_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS
        if steady:
            roof["step_frac_of_hbm_peak_steady_state"] = step_bytes / (steady["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS

    variants = None
    if rank == 0 and world == 1 and not a.no_variants and a.variant == "p2" and a.size == 512:
        net.close()
        del frames, recon
        torch.cuda.empty_cache()
        M4, M3, M5 = [8, 16, 32, 64], [8, 16, 32], [8, 16, 32, 64, 128]
        variants = {
            "per_frame_form": fft_variant(aefft, torch, np, ctx, "cfg3-P2, per-frame form (NOOPFORM: every layer for every frame, round 1's step)",
                                          512, M4, 2, 32, steps=100, warmup=20, flags=("NOOPFORM",)),
            "u8_frames": fft_variant(aefft, torch, np, ctx, "cfg3-P2 with the frames resident as 8-bit pixels (aefft_net_step_grad_u8: what a camera delivers; the headline "
                                     "keeps the reference's float frames); same pixel values, same results", 512, M4, 2, 32, steps=200, warmup=20, u8=True),
            "p1": fft_variant(aefft, torch, np, ctx, "cfg3-P1: as the headline but pool 1/layer (all pairs at 512x512; 5.7 GB of kernel spectra)",
                              512, M4, 1, 32, steps=30, warmup=5),
            "cfg2": fft_variant(aefft, torch, np, ctx, "cfg2: 256x256x3, 3 pairs 3->8->16->32, 5x5, pool 2/layer, B = 1 (BASELINE configs[1])",
                                256, M3, 2, 1, steps=500, warmup=50),
            "cfg5": fft_variant(aefft, torch, np, ctx, "cfg5 per-GPU shape: 1024x1024x3, 5 pairs 3->8->...->128, 5x5, pool 2/layer, tied weights + "
                                "multiobjective (sym=1, maxdiff=1), 32 frames (BASELINE configs[4])", 1024, M5, 2, 32, steps=60, warmup=10, sym=1, maxdiff=1),
            "spatial": variant_spatial(aefft, torch, np, ctx)}

    # (the CPU baseline last: its 16 worker processes leave the host's CPU share throttled for a moment, which the launch-bound variants would see)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(N, s if s > 1 else 2)

    if rank == 0:
        out = {
            "metric": "frames/s fwd+bwd (FFT mode, 512x512, 4 layers)", "value": (world * B * a.steps / dt) if mse_finite else None, "unit": "frames/s",
            "n_gpus": world, "world_size": (dist.get_world_size() if dist is not None else 1), "steps": a.steps, "warmup": a.warmup, "preheat_steps": a.preheat, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3-{a.variant.upper()}: {N}x{N}x3 frames, 4 pairs 3->8->16->32->64, 5x5, pool {s}/layer, FFT mode, "
                                   f"fwd + 1 loop-body iteration per pair ({'per-frame form' if 'NOOPFORM' in a.flags else 'operator form'}); every step produces "
                                   "reconstructions, batch-mean gradients, updated weights and the per-pair post-update MSE partial sums -- the sums themselves are "
                                   "formed by the next step's gradient launch (no per-step MSE read-back inside the timed loop)", "frames_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"dp{world}" + (" (RCCL all-reduce of packed kernel-support gradients)" if world > 1 else "")},
            "mse_per_pair": mse_host, "mse_first_step": mse_first, "mse_finite": mse_finite,
        }
        if not mse_finite:
            out["invalid_reason"] = "non-finite post-update MSE at the end of the timed region: the measurement is void (value = null)"
        if steady:
            out["steady_state"] = steady
        if dp_diag:
            out["data_parallel"] = dp_diag
        if roof:
            out["roofline"] = roof
        if cpu:
            out["cpu_baseline"] = cpu
        if variants:
            out["variants"] = variants
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(finite_json(out)) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
