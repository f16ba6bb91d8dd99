"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the CPU-path checkers.

`port()`      -> oracle/libcpu_ref.so   (oracle/cpu_ref.c, the plain-C restatement)
`reference()` -> oracle/_ref/libnetlib_ref.so (the reference's own Conv/backprop/Pool/Portion,
                 compiled from /root/reference by oracle/Makefile; None when not built)

Both expose the same four calls on flat float32 numpy arrays (see cpu_ref.c for layouts):
    conv(in, c, b) -> out ; backprop(in, out, hin, c, b, f, p, del) -> (c, b, f, p)
    pool(in, out_shape, scale) -> out ; portion(in, q) -> in_s
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FP = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(_FP)


def build():
    """Compile the checkers (gcc; plus oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


class _Lib:
    def __init__(self, path, prefix, kind):
        self.kind = kind
        self.lib = C.CDLL(path)
        self.pfx = prefix
        for name in ("conv", "backprop", "pool", "portion"):
            getattr(self.lib, prefix + name).restype = None
        if prefix == "cpu_ref_":
            self.lib.cpu_ref_backprop.restype = C.c_float

    def conv(self, x, c, b):
        x = np.ascontiguousarray(x, np.float32); c = np.ascontiguousarray(c, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        dD, Nx, Ny = x.shape
        dM, dD2, Nk, Nl = c.shape
        assert dD2 == dD and b.shape == (dM,)
        out = np.zeros((dM, Nx, Ny), np.float32)
        getattr(self.lib, self.pfx + "conv")(_p(x), _p(out), _p(c), _p(b), dD, dM, Nx, Ny, Nk, Nl)
        return out

    def backprop(self, x, out, hin, c, b, f, p, dele):
        x = np.ascontiguousarray(x, np.float32); out = np.ascontiguousarray(out, np.float32)
        hin = np.ascontiguousarray(hin, np.float32)
        c = np.array(c, np.float32, order="C"); f = np.array(f, np.float32, order="C")
        b = np.array(b, np.float32); p = np.array(p, np.float32)
        dD, Nx, Ny = x.shape
        dM, dD2, Nk, Nl = c.shape
        assert dD2 == dD and f.shape == (dD, dM, Nk, Nl) and hin.shape == (dM, Nx, Ny) and out.shape == x.shape
        getattr(self.lib, self.pfx + "backprop")(_p(x), _p(out), _p(hin), _p(c), _p(b), _p(f), _p(p),
                                                C.c_float(dele), dD, dM, Nx, Ny, Nk, Nl)
        return c, b, f, p

    def pool(self, x, out_shape, scale, out_init=None):
        x = np.ascontiguousarray(x, np.float32)
        D, Nxi, Nyi = x.shape
        out = np.zeros(out_shape, np.float32) if out_init is None else np.array(out_init, np.float32)
        getattr(self.lib, self.pfx + "pool")(_p(x), _p(out), D, Nxi, Nyi, out.shape[1], out.shape[2], int(scale))
        return out

    def portion(self, x, q):
        x = np.ascontiguousarray(x, np.float32)
        ch, Nx, Ny = x.shape
        out = np.zeros((ch, Nx // q, Ny // q), np.float32)
        getattr(self.lib, self.pfx + "portion")(_p(x), _p(out), ch, Nx, Ny, int(q))
        return out


def port():
    path = os.path.join(_HERE, "libcpu_ref.so")
    if not os.path.exists(path):
        build()
    return _Lib(path, "cpu_ref_", "port")


def reference():
    path = os.path.join(_HERE, "_ref", "libnetlib_ref.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/source/netlib.cpp"):
            build()
        else:
            return None
    return _Lib(path, "ref_", "reference")


def time_pair_frame(args):
    """bench.py's cpu_baseline worker (runs in a spawned process, one per core): ONE frame through the CPU path of one
    encoder/decoder pair -- Pool -> Conv -> Conv -> Pool(-s) -> backprop (autoencoder.cpp:135-150,200) -- returns seconds."""
    import time
    N, scale, dD, dM, Nk, seed = args
    L = reference() or port()
    rng = np.random.default_rng(seed)
    n = N // scale
    x = np.floor(rng.uniform(0, 256, (dD, N, N))).astype(np.float32)
    c = rng.uniform(-3, 3, (dM, dD, Nk, Nk)).astype(np.float32); f = rng.uniform(-3, 3, (dD, dM, Nk, Nk)).astype(np.float32)
    b = rng.uniform(-3, 3, dM).astype(np.float32); p = rng.uniform(-3, 3, dD).astype(np.float32)
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)   # the reference prints "mse: ..."
    try:
        t0 = time.perf_counter()
        pin = L.pool(x, (dD, n, n), scale)
        h = L.conv(pin, c, b)
        o = L.conv(h, f, p)
        L.pool(o, (dD, N, N), -scale)
        L.backprop(pin, o, h, c, b, f, p, 0.2)
        dt = time.perf_counter() - t0
    finally:
        os.dup2(saved, 1); os.close(devnull); os.close(saved)
    return dt, L.kind
