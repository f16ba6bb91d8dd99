"""Data-parallel host logic (no reference counterpart -- the reference is single-GPU).

Frames shard across ranks (one process per GPU, torch.distributed: backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  Forward, error and the frequency-space gradient are independent
per frame; C2R and shrink are linear, so every rank reduces its own frames to the kernel-support-sized
packed buffer  [dck | dfk | db | dp] per pair  (aefft_net_grad_buffer), ONE all-reduce(SUM) of that buffer
(0.54 MB for the 4-pair 512x512 network) is followed by the identical clipped-momentum update on every
rank with grad_scale = 1/world.  Weights stay replicated bit-for-bit because every rank applies the same
float32 update to the same all-reduced buffer.

The MSE rides in the same message (SURVEY 8e "+ 1 float MSE"): the packed buffer ends in L floats, the post-update MSE per pair of
this rank's shard as the PREVIOUS step left it (aefft_net_step_apply writes them there; the post-update MSE of a step needs that
step's reduced gradients, so it cannot travel with them).  After the all-reduce that tail holds the sum over ranks; mse_tail() turns
it into the global-batch mean (one step behind), and flush_mse() reduces the last step's values with one more small collective."""
import numpy as np


def grad_layout(dims):
    """Offsets (in floats) of each pair's segments inside the packed gradient buffer; mirrors
    aefft_net_create (aefft_capi.hip: goff += 2*nk + dM + dD).  dims: list of dict(dM, dD, Nk, Nl).
    Returns (layout, n): n = floats of the gradient part; the buffer the library exposes is n + len(dims) long (mse_tail_slice)."""
    out, off = [], 0
    for g in dims:
        nk = g["dM"] * g["dD"] * g["Nk"] * g["Nl"]
        out.append(dict(dck=(off, nk), dfk=(off + nk, nk), db=(off + 2 * nk, g["dM"]), dp=(off + 2 * nk + g["dM"], g["dD"])))
        off += 2 * nk + g["dM"] + g["dD"]
    return out, off


def mse_tail_slice(dims):
    """Where the per-pair post-update MSEs of the previous step sit in the packed buffer (aefft_net_grad_buffer: nfloats = gradients + L)."""
    _, n = grad_layout(dims)
    return slice(n, n + len(dims))


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
        self.tail = mse_tail_slice(net.dims)
        assert self.gbuf.numel() == self.tail.stop, "packed buffer: gradients + one MSE float per pair"
        self._mse_prev = None            # global-batch MSE of the step before the last __call__ (device tensor, stream-ordered)
        self._scale = 1.0
        self.timers = None               # set to a list to collect (grad_ms, allreduce_ms, apply_ms) per step (events on the library's stream)

    def __call__(self, frames, recon, del0, maxdiff=0, sym=0, mse=None):
        import torch
        st = self.net.ctx.torch_stream()
        ev = None
        if self.timers is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(st)
        self.net.step_grad(frames, recon)
        # the collective is enqueued on the library's own stream: gradients -> all-reduce -> update stay ordered
        # without any host synchronisation
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            with torch.cuda.stream(st):
                if ev: ev[1].record(st)
                scale = allreduce_sum_(self.gbuf, self.group)
                if ev: ev[2].record(st)
                # the tail now holds the sum over ranks of the previous step's post-update MSEs: aefft_net_step_apply itself saves the
                # global mean behind the buffer before its own values take the tail (nothing is enqueued here: a kernel between the
                # collective and the update half would sit on the step's critical path)
                self._mse_prev = self.net.mse_prev_global()
        else:
            # no process group: nothing is enqueued between the two halves (a kernel there -- even the 4-float copy of the tail --
            # is a launch on the step's critical path); the step's own MSEs are the global ones (`mse` of aefft_net_step_apply)
            scale = 1.0
            if ev: ev[1].record(st); ev[2].record(st)
            self._mse_prev = self.net.mse_prev_global()
        self._scale = scale
        self.net.step_apply(del0, maxdiff, sym, scale, mse)
        if ev:
            ev[3].record(st)
            self.timers.append(ev)

    def mse_tail(self):
        """global-batch post-update MSE per pair of the step BEFORE the last one (it travelled in the last gradient all-reduce and was saved
        behind the packed buffer by the last aefft_net_step_apply): a stream-ordered view, valid until the next step; zeros after the first step"""
        return self._mse_prev

    def flush_mse(self):
        """global-batch post-update MSE per pair of the LAST step: one small all-reduce of the buffer's tail (end of training / logging points)"""
        import torch
        self.net.last_mse()              # (a step_apply without an mse output leaves the sums to the next step: form them now)
        with torch.cuda.stream(self.net.ctx.torch_stream()):
            t = self.gbuf[self.tail].clone()
            scale = allreduce_sum_(t, self.group)
            return t * scale

    def phase_ms(self):
        """mean (grad half, all-reduce, apply half) milliseconds over the steps recorded in self.timers (synchronises)"""
        import torch
        torch.cuda.synchronize()
        n = max(len(self.timers), 1)
        return [sum(e[i].elapsed_time(e[i + 1]) for e in self.timers) / n for i in range(3)]

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


class RcclStep:
    """The same step with NOTHING of Python or torch.distributed between its two halves: libaefft_dp.so (include/aefft_dp.h) enqueues
    step_grad -> ncclAllReduce on the library's stream -> step_apply(1/world) in one C call.  The ncclUniqueId reaches the other ranks
    through `bcast` (a callable taking and returning a uint8 torch tensor of DP_ID_BYTES: e.g. a torch.distributed.broadcast from rank 0
    on whatever group the host already has); world = 1 needs none."""

    def __init__(self, net, rank=0, world=1, bcast=None):
        import ctypes as C
        import importlib
        import torch
        aefft = importlib.import_module(__package__)
        self.L = aefft.dp_lib()
        self.net, self.rank, self.world = net, rank, world
        idbuf = torch.zeros(aefft.DP_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            rc = self.L.aefft_dp_unique_id(C.c_void_p(idbuf.data_ptr()))
            if rc != 0:
                raise aefft.AefftError(f"aefft_dp_unique_id failed with code {rc}")
        if world > 1:
            assert bcast is not None, "world > 1: the unique id must be broadcast from rank 0"
            idbuf = bcast(idbuf).cpu().contiguous()
        h = C.c_void_p()
        rc = self.L.aefft_dp_create(net.h, net.ctx.h, rank, world, C.c_void_p(idbuf.data_ptr()), C.byref(h))
        if rc != 0:
            raise aefft.AefftError(f"aefft_dp_create failed with code {rc}")
        self.h = h
        self._err = aefft.AefftError

    def _check(self, rc):
        if rc != 0:
            raise self._err(f"aefft_dp error {rc}: {self.L.aefft_dp_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.aefft_dp_destroy(self.h)
            self.h = None

    def __call__(self, frames, recon, del0, maxdiff=0, sym=0, mse=None):
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.L.aefft_dp_step(self.h, p(frames), p(recon), del0, maxdiff, sym, p(mse)))

    def run(self, frames, recon, del0, nsteps, maxdiff=0, sym=0):
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(self.L.aefft_dp_run(self.h, p(frames), p(recon), del0, maxdiff, sym, int(nsteps)))

    def profile(self, frames, recon, del0, nsteps, maxdiff=0, sym=0):
        """mean (grad half, all-reduce, update half) ms from events on the library's stream, and the host's microseconds per step"""
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        ph = (C.c_double * 3)(); host = C.c_double()
        self._check(self.L.aefft_dp_profile(self.h, p(frames), p(recon), del0, maxdiff, sym, int(nsteps), ph, C.byref(host)))
        return list(ph), host.value

    def flush_mse(self):
        import ctypes as C
        out = np.zeros(self.net.npairs, np.float32)
        self._check(self.L.aefft_dp_flush_mse(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def allreduce_bytes(self):
        return int(self.L.aefft_dp_allreduce_bytes(self.h))

    def replicas_agree(self):
        r = self.L.aefft_dp_replicas_agree(self.h)
        if r < 0:
            raise self._err(f"aefft_dp_replicas_agree failed with code {r}")
        return bool(r)
