"""Data-parallel host logic (no reference counterpart -- the reference is single-GPU).

Frames shard across ranks (one process per GPU, torch.distributed: backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  Forward, error and the frequency-space gradient are independent
per frame; C2R and shrink are linear, so every rank reduces its own frames to the kernel-support-sized
packed buffer  [dck | dfk | db | dp] per pair  (aefft_net_grad_buffer), ONE all-reduce(SUM) of that buffer
(0.54 MB for the 4-pair 512x512 network) is followed by the identical clipped-momentum update on every
rank with grad_scale = 1/world.  Weights stay replicated bit-for-bit because every rank applies the same
float32 update to the same all-reduced buffer."""
import numpy as np


def grad_layout(dims):
    """Offsets (in floats) of each pair's segments inside the packed gradient buffer; mirrors
    aefft_net_create (aefft_capi.hip: goff += 2*nk + dM + dD).  dims: list of dict(dM, dD, Nk, Nl)."""
    out, off = [], 0
    for g in dims:
        nk = g["dM"] * g["dD"] * g["Nk"] * g["Nl"]
        out.append(dict(dck=(off, nk), dfk=(off + nk, nk), db=(off + 2 * nk, g["dM"]), dp=(off + 2 * nk + g["dM"], g["dD"])))
        off += 2 * nk + g["dM"] + g["dD"]
    return out, off


def pack_grads(per_pair, dims):
    """per_pair: list of (dck, dfk, db, dp) arrays -> one float32 vector in the C layout."""
    lay, n = grad_layout(dims)
    buf = np.zeros(n, np.float32)
    for (dck, dfk, db, dp), L in zip(per_pair, lay):
        for a, k in ((dck, "dck"), (dfk, "dfk"), (db, "db"), (dp, "dp")):
            o, m = L[k]
            buf[o:o + m] = np.asarray(a, np.float32).ravel()
    return buf


def unpack_grads(buf, dims):
    lay, _ = grad_layout(dims)
    out = []
    for g, L in zip(dims, lay):
        shp = dict(dck=(g["dM"], g["dD"], g["Nk"], g["Nl"]), dfk=(g["dD"], g["dM"], g["Nk"], g["Nl"]), db=(g["dM"],), dp=(g["dD"],))
        out.append(tuple(np.asarray(buf[L[k][0]:L[k][0] + L[k][1]]).reshape(shp[k]) for k in ("dck", "dfk", "db", "dp")))
    return out


def allreduce_sum_(tensor, group=None):
    """In-place SUM all-reduce of the packed buffer; returns the scale (1/world) the update must apply."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 1.0                         # (an initialised group of ONE rank still runs the collective: same code path as N > 1)
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


class DataParallelStep:
    """One training step over a sharded batch: step_grad -> all-reduce -> step_apply on every rank."""

    def __init__(self, net, group=None):
        self.net, self.group = net, group
        self.gbuf = net.grad_buffer()

    def __call__(self, frames, recon, del0, maxdiff=0, sym=0, mse=None):
        self.net.step_grad(frames, recon)
        # the collective is enqueued on the library's own stream: gradients -> all-reduce -> update stay ordered
        # without any host synchronisation
        import torch
        with torch.cuda.stream(self.net.ctx.torch_stream()):
            scale = allreduce_sum_(self.gbuf, self.group)
        self.net.step_apply(del0, maxdiff, sym, scale, mse)

    def replicas_agree(self):
        """All ranks hold identical weights (checksum of every pair's tensors, all-gathered)."""
        import torch
        import torch.distributed as dist
        sums = []
        for l in range(self.net.npairs):
            sums += [float(np.float64(a.astype(np.float64).sum())) for a in self.net.get_pair(l)]
        t = torch.tensor(sums, dtype=torch.float64)
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return True
        dev = self.gbuf.device if self.gbuf.is_cuda else "cpu"
        t = t.to(dev)
        gathered = [torch.empty_like(t) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(gathered, t, group=self.group)
        return all(bool((g == gathered[0]).all()) for g in gathered)
