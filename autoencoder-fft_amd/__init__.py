"""ctypes binding of libaefft.so (include/aefft.h) -- plumbing for tests and bench.py.

The product is the HIP library; this module only moves pointers.  torch provides device
memory (tensor.data_ptr()), the stream and torch.distributed; no arithmetic happens here.
There is NO CPU fallback: importing succeeds without the library only so that CPU-side tests
can inspect the host logic, but every compute call raises if libaefft.so or the GPU is missing.

Import with:  aefft = importlib.import_module("autoencoder-fft_amd")
"""
import atexit
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AEFFT_LIB: development only -- tools/x*.sh point the binding at an experiment build under build_x/ instead of copying it over the product)
LIB_PATH = os.environ.get("AEFFT_LIB") or os.path.join(_HERE, "libaefft.so")

OK, EINVAL, EHIP, ENOMEM, ESTATE = 0, 1, 2, 3, 4


# include/aefft.h AEFFT_F_* (development switches; tests/test_abi.py checks this table against the header)
FLAGS = {n: 1 << i for i, n in enumerate(
    ["NOLAZY", "NOCOMPACT", "NOQPATH", "NOFUSEMSE", "NOGROUP", "NOMFMA", "NOGFWD", "NOOVERLAP", "NOFUSECROP", "GTAPS",
     "NOPREFETCH", "NODEFER", "NOTILEDSPATIAL", "NOFAST", "NOSPLITK", "POISON", "NOOPFORM", "NOCHAIN", "NOFUSEUPD", "NOAHEAD", "NORCORR", "NOLAZYMSE", "SMALLOVERLAP", "CHAINMSE"])}


class AefftError(RuntimeError):
    pass


class NetDesc(C.Structure):
    _fields_ = [("D", C.c_int), ("Nx", C.c_int), ("Ny", C.c_int), ("npairs", C.c_int),
                ("maps", C.POINTER(C.c_int)), ("Nk", C.POINTER(C.c_int)), ("Nl", C.POINTER(C.c_int)),
                ("scale", C.POINTER(C.c_int)), ("batch", C.c_int)]


_lib = None
_vp, _fp, _i, _l, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float   # device float* passed as void*

# name -> (restype, argtypes); must list every symbol declared in include/aefft.h
SIGNATURES = {
    "aefft_ctx_create": (_i, [C.POINTER(_vp), _i, _vp, _i]),
    "aefft_ctx_destroy": (None, [_vp]),
    "aefft_last_error": (C.c_char_p, [_vp]),
    "aefft_sync": (_i, [_vp]),
    "aefft_ctx_partition": (_i, [_vp, _i]),
    "aefft_ctx_set_flags": (_i, [_vp, C.c_uint]),
    "aefft_ctx_get_flags": (C.c_uint, [_vp]),
    "aefft_stream": (_vp, [_vp]),
    "aefft_version": (C.c_char_p, []),
    "aefft_r2c": (_i, [_vp, _fp, _fp, _l, _i, _i]),
    "aefft_c2r": (_i, [_vp, _fp, _fp, _l, _i, _i, _f]),
    "aefft_pool": (_i, [_vp, _fp, _fp, _l, _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "aefft_r2c_pool": (_i, [_vp, _fp, _fp, _l, _i, _i, _i]),
    "aefft_unpool_c2r": (_i, [_vp, _fp, _fp, _l, _i, _i, _i, _f]),
    "aefft_kernel_spectrum": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _i]),
    "aefft_kernel_export": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _i, _i]),
    "aefft_conv": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i]),
    "aefft_gradient": (_i, [_vp] + [_fp] * 10 + [_i] * 5),
    "aefft_mse": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i]),
    "aefft_update": (_i, [_vp] + [_fp] * 14 + [_i] * 6 + [_f, _i]),
    "aefft_conv_spatial": (_i, [_vp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i]),
    "aefft_pool_spatial": (_i, [_vp, _fp, _fp, C.c_long, _i, _i, _i, _i, _i]),
    "aefft_pool_conv_spatial": (_i, [_vp, _fp, _fp, _fp, _fp, _fp] + [_i] * 9),
    "aefft_backprop_spatial": (_i, [_vp] + [_fp] * 15 + [_i] * 7 + [_f, _f, _i, _i]),
    "aefft_step_spatial": (_i, [_vp] + [_fp] * 15 + [_i] * 7 + [_f, _f, _i, _i]),
    "aefft_net_create": (_i, [_vp, C.POINTER(NetDesc), C.POINTER(_vp)]),
    "aefft_net_destroy": (None, [_vp]),
    "aefft_net_npairs": (_i, [_vp]),
    "aefft_net_pair_shape": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "aefft_net_set_pair": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "aefft_net_get_pair": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "aefft_net_pair_spectra": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "aefft_net_load_spectra": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "aefft_net_store_spectra": (_i, [_vp, _i, _vp, _vp]),
    "aefft_net_forward": (_i, [_vp, _fp, _fp]),
    "aefft_net_get_layer": (_i, [_vp, _i, _fp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "aefft_net_layers_layout": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "aefft_net_get_layers": (_i, [_vp, _fp]),
    "aefft_magnitude": (_i, [_vp, _fp, _fp, _l, _i, _i, _i, _i]),
    "aefft_net_train_pair": (_i, [_vp, _i, _i, _f, _i, _i, _vp]),
    "aefft_net_step_grad": (_i, [_vp, _fp, _fp]),
    "aefft_net_step_grad_u8": (_i, [_vp, _vp, _fp]),
    "aefft_net_forward_u8": (_i, [_vp, _vp, _fp]),
    "aefft_net_set_input_ready": (_i, [_vp, _i]),
    "aefft_net_grad_buffer": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "aefft_net_step_form": (_i, [_vp]),
    "aefft_net_last_mse": (_i, [_vp, _vp]),
    "aefft_net_step_apply": (_i, [_vp, _f, _i, _i, _f, _fp]),
    "aefft_net_reset_momentum": (_i, [_vp]),
    "aefft_prof_enable": (_i, [_vp, _i]),
    "aefft_prof_count": (_i, []),
    "aefft_prof_name": (C.c_char_p, [_i]),
    "aefft_prof_read": (_i, [_vp, _i, C.POINTER(_l), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "aefft_prof_reset": (_i, [_vp]),
}


def lib():
    """Load libaefft.so and declare every prototype; raises if the HIP library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AefftError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


DP_LIB_PATH = os.path.join(_HERE, "libaefft_dp.so")
_dp_lib = None
# include/aefft_dp.h (libaefft_dp.so = libaefft.so + librccl): name -> (restype, argtypes)
DP_SIGNATURES = {
    "aefft_dp_unique_id": (_i, [_vp]),
    "aefft_dp_create": (_i, [_vp, _vp, _i, _i, _vp, C.POINTER(_vp)]),
    "aefft_dp_destroy": (None, [_vp]),
    "aefft_dp_last_error": (C.c_char_p, [_vp]),
    "aefft_dp_step": (_i, [_vp, _fp, _fp, _f, _i, _i, _fp]),
    "aefft_dp_run": (_i, [_vp, _fp, _fp, _f, _i, _i, _i]),
    "aefft_dp_profile": (_i, [_vp, _fp, _fp, _f, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "aefft_dp_flush_mse": (_i, [_vp, _vp]),
    "aefft_dp_allreduce_bytes": (C.c_size_t, [_vp]),
    "aefft_dp_replicas_agree": (_i, [_vp]),
}
DP_ID_BYTES = 128


def dp_lib():
    """Load libaefft_dp.so (the data-parallel step over RCCL, include/aefft_dp.h); raises if it was not built."""
    global _dp_lib
    if _dp_lib is None:
        lib()                                             # libaefft.so first: the same image the nets were created from
        if not os.path.exists(DP_LIB_PATH):
            raise AefftError(f"{DP_LIB_PATH} not built: run `make -C autoencoder-fft_amd/csrc dp` (__graft_entry__.build does)")
        L = C.CDLL(DP_LIB_PATH)
        for name, (res, args) in DP_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _dp_lib = L
    return _dp_lib


_exiting = False


def _mark_exit():
    global _exiting
    _exiting = True


atexit.register(_mark_exit)


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _is_u8(t):
    import torch
    return t is not None and t.dtype == torch.uint8


def _hptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """aefft_ctx on torch's current device, enqueuing on torch's current stream by default."""

    def __init__(self, device=None, use_torch_stream=True):
        import torch
        if not torch.cuda.is_available():
            raise AefftError("no MI355X visible (torch.cuda.is_available() is False); aefft has no CPU path")
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        torch.cuda.set_device(self.device)
        self.L = lib()
        h = C.c_void_p()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if use_torch_stream else None
        rc = self.L.aefft_ctx_create(C.byref(h), self.device, stream, 0 if use_torch_stream else 1)
        if rc != OK:
            raise AefftError(f"aefft_ctx_create failed with code {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.aefft_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        if _exiting:      # interpreter shutdown: the HIP runtime may already be gone; the process frees everything
            return
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != OK:
            raise AefftError(f"aefft error {rc}: {self.L.aefft_last_error(self.h).decode()}")

    def sync(self):
        self.check(self.L.aefft_sync(self.h))

    def partition(self, side_cus):
        """aefft_ctx_partition: CU-masked streams -- the side streams get `side_cus` compute units, the context stream the rest"""
        self.check(self.L.aefft_ctx_partition(self.h, int(side_cus)))

    def torch_stream(self):
        """The library's stream as a torch stream object (so collectives / torch ops can be ordered on it)."""
        ptr = self.L.aefft_stream(self.h)
        if not ptr:
            return self.torch.cuda.default_stream(self.device)
        return self.torch.cuda.ExternalStream(ptr, device=f"cuda:{self.device}")

    # ---- helpers on torch tensors (float32 / complex64, contiguous, on this device) ----
    def empty(self, *shape, dtype=None):
        t = self.torch
        return t.empty(*shape, dtype=dtype or t.float32, device=f"cuda:{self.device}")

    def dev(self, a, dtype=None):
        t = self.torch
        a = np.ascontiguousarray(a)
        if np.iscomplexobj(a):
            a = a.astype(np.complex64)
        else:
            a = a.astype(np.float32)
        return t.from_numpy(a).to(f"cuda:{self.device}")

    # ---- op level ----
    def r2c(self, x):
        *lead, Nx, Ny = x.shape
        planes = int(np.prod(lead)) if lead else 1
        X = self.empty(*lead, Nx, Ny // 2 + 1, dtype=self.torch.complex64)
        self.check(self.L.aefft_r2c(self.h, _ptr(x), _ptr(X), planes, Nx, Ny))
        return X

    def c2r(self, X, Ny, scale=None):
        *lead, Nx, Nyr = X.shape
        planes = int(np.prod(lead)) if lead else 1
        x = self.empty(*lead, Nx, Ny)
        s = 1.0 / (Nx * Ny) if scale is None else scale
        self.check(self.L.aefft_c2r(self.h, _ptr(X), _ptr(x), planes, Nx, Ny, s))
        return x

    def pool(self, X, Ny, scale):
        *lead, Nx, Nyr = X.shape
        planes = int(np.prod(lead)) if lead else 1
        a = abs(scale)
        nx, ny = (Nx // a, Ny // a) if scale > 0 else (Nx * a, Ny * a)
        Xs = self.empty(*lead, nx, ny // 2 + 1, dtype=self.torch.complex64)
        onx, ony = C.c_int(), C.c_int()
        self.check(self.L.aefft_pool(self.h, _ptr(X), _ptr(Xs), planes, Nx, Ny, scale, C.byref(onx), C.byref(ony)))
        assert (onx.value, ony.value) == (nx, ny)
        return Xs, nx, ny

    def r2c_pool(self, x, scale):
        *lead, Nx, Ny = x.shape
        planes = int(np.prod(lead)) if lead else 1
        Xs = self.empty(*lead, Nx // scale, Ny // scale // 2 + 1, dtype=self.torch.complex64)
        self.check(self.L.aefft_r2c_pool(self.h, _ptr(x), _ptr(Xs), planes, Nx, Ny, scale))
        return Xs

    def unpool_c2r(self, Xs, Nys, scale, out_scale):
        *lead, Nxs, _ = Xs.shape
        planes = int(np.prod(lead)) if lead else 1
        a = -scale
        x = self.empty(*lead, Nxs * a, Nys * a)
        self.check(self.L.aefft_unpool_c2r(self.h, _ptr(Xs), _ptr(x), planes, Nxs, Nys, scale, out_scale))
        return x

    def kernel_spectrum(self, k, Nx, Ny):
        nA, nB, Nk, Nl = k.shape
        K = self.empty(nA, nB, Nx, Ny // 2 + 1, dtype=self.torch.complex64)
        self.check(self.L.aefft_kernel_spectrum(self.h, _ptr(k), _ptr(K), nA, nB, Nk, Nl, Nx, Ny))
        return K

    def kernel_export(self, K, Nk, Nl, Ny):
        nA, nB, Nx, _ = K.shape
        k = self.empty(nA, nB, Nk, Nl)
        self.check(self.L.aefft_kernel_export(self.h, _ptr(K), _ptr(k), nA, nB, Nk, Nl, Nx, Ny))
        return k

    def conv(self, X, Cs, bias, Ny):
        B, dD, Nx, _ = X.shape
        dM = Cs.shape[0]
        O = self.empty(B, dM, Nx, Ny // 2 + 1, dtype=self.torch.complex64)
        self.check(self.L.aefft_conv(self.h, _ptr(X), _ptr(Cs), _ptr(bias), _ptr(O), B, dM, dD, Nx, Ny))
        return O

    def gradient(self, Xin, Xout, O, Cs, Fs, b, Ny):
        B, dD, Nx, Nyr = Xin.shape
        dM = Cs.shape[0]
        c64 = self.torch.complex64
        dc = self.empty(dM, dD, Nx, Nyr, dtype=c64); df = self.empty(dD, dM, Nx, Nyr, dtype=c64)
        db = self.empty(dM); dp = self.empty(dD)
        self.check(self.L.aefft_gradient(self.h, _ptr(Xin), _ptr(Xout), _ptr(O), _ptr(Cs), _ptr(Fs), _ptr(b),
                                         _ptr(dc), _ptr(df), _ptr(db), _ptr(dp), B, dM, dD, Nx, Ny))
        return dc, df, db, dp

    def mse(self, T, O, dM, Ny):
        B, dD, Nx, _ = T.shape
        out = self.empty(1)
        self.check(self.L.aefft_mse(self.h, _ptr(T), _ptr(O), _ptr(out), B, dM, dD, Nx, Ny))
        return out

    def update(self, c, f, b, p, Cs, Fs, dc, df, db, dp, Dc, Df, Db, Dp, Ny, dele, maxdiff):
        dM, dD, Nk, Nl = c.shape
        Nx = Cs.shape[2]
        self.check(self.L.aefft_update(self.h, _ptr(c), _ptr(f), _ptr(b), _ptr(p), _ptr(Cs), _ptr(Fs), _ptr(dc), _ptr(df),
                                       _ptr(db), _ptr(dp), _ptr(Dc), _ptr(Df), _ptr(Db), _ptr(Dp),
                                       dM, dD, Nx, Ny, Nk, Nl, dele, maxdiff))

    def magnitude(self, X, Ny, ch, shift=False):
        *lead, Nx, Nyr = X.shape
        planes = int(np.prod(lead)) if lead else 1
        mag = self.empty(*lead, Nx, Ny)
        self.check(self.L.aefft_magnitude(self.h, _ptr(X), _ptr(mag), planes, ch, Nx, Ny, 1 if shift else 0))
        return mag

    def conv_spatial(self, x, c, b, semantics="gpu"):
        B, dD, Nx, Ny = x.shape
        dM, _, Nk, Nl = c.shape
        out = self.empty(B, dM, Nx, Ny)
        self.check(self.L.aefft_conv_spatial(self.h, _ptr(x), _ptr(out), _ptr(c), _ptr(b), B, dD, dM, Nx, Ny, Nk, Nl,
                                             0 if semantics == "gpu" else 1))
        return out

    def pool_spatial(self, x, out_shape, scale):
        """netlib.cpp Pool on the device: x [..., Nxi, Nyi] -> zeros(out_shape) filled where the reference writes."""
        import torch
        out = torch.zeros(tuple(x.shape[:-2]) + tuple(out_shape), dtype=torch.float32, device=x.device)
        planes = int(np.prod(x.shape[:-2]))
        self.check(self.L.aefft_pool_spatial(self.h, _ptr(x), _ptr(out), planes, x.shape[-2], x.shape[-1], out_shape[0], out_shape[1], scale))
        return out

    def pool_conv_spatial(self, x, c, b, scale, semantics="gpu", want_pooled=True):
        """Pool(scale) + Conv_gpu in one launch; returns (pooled layer or None, conv output)."""
        B, dD, Nxi, Nyi = x.shape
        dM, _, Nk, Nl = c.shape
        Nx, Ny = Nxi // scale, Nyi // scale
        pooled = self.empty(B, dD, Nx, Ny) if want_pooled else None
        out = self.empty(B, dM, Nx, Ny)
        self.check(self.L.aefft_pool_conv_spatial(self.h, _ptr(x), _ptr(pooled), _ptr(out), _ptr(c), _ptr(b), B, dD, dM, Nx, Ny, scale, Nk, Nl,
                                                  0 if semantics == "gpu" else 1))
        return pooled, out

    def backprop_spatial(self, x, out, hin, c, b, f, p, mom, grads, delmax, alpha, tied=False, semantics="gpu"):
        """mom = (dc, db, df, dp), grads = (ddc, ddb, ddf, ddp): torch tensors updated in place."""
        B, dD, Nx, Ny = x.shape
        dM, _, Nk, Nl = c.shape
        dc, db, df, dp = mom
        ddc, ddb, ddf, ddp = grads
        self.check(self.L.aefft_backprop_spatial(self.h, _ptr(x), _ptr(out), _ptr(hin), _ptr(c), _ptr(b), _ptr(f), _ptr(p),
                                                 _ptr(dc), _ptr(db), _ptr(df), _ptr(dp), _ptr(ddc), _ptr(ddb), _ptr(ddf), _ptr(ddp),
                                                 B, dD, dM, Nx, Ny, Nk, Nl, delmax, alpha, 1 if tied else 0,
                                                 {"gpu": 0, "cpu": 1, "cuda_compat": 2}[semantics]))

    def step_spatial(self, x, c, b, f, p, mom, grads, delmax, alpha, tied=False, semantics="gpu", hin=None, out=None):
        """Conv_gpu + Conv_gpu + backprop_gpu[_cc] in one call (aefft_step_spatial); returns (hin, out)."""
        B, dD, Nx, Ny = x.shape
        dM, _, Nk, Nl = c.shape
        hin = self.empty(B, dM, Nx, Ny) if hin is None else hin
        out = self.empty(B, dD, Nx, Ny) if out is None else out
        dc, db, df, dp = mom
        ddc, ddb, ddf, ddp = grads
        self.check(self.L.aefft_step_spatial(self.h, _ptr(x), _ptr(hin), _ptr(out), _ptr(c), _ptr(b), _ptr(f), _ptr(p),
                                             _ptr(dc), _ptr(db), _ptr(df), _ptr(dp), _ptr(ddc), _ptr(ddb), _ptr(ddf), _ptr(ddp),
                                             B, dD, dM, Nx, Ny, Nk, Nl, delmax, alpha, 1 if tied else 0, {"gpu": 0, "cpu": 1}[semantics]))
        return hin, out

    def set_flags(self, *names):
        """Development switches (include/aefft.h AEFFT_F_*), by name without the prefix; no names = defaults."""
        v = 0
        for nme in names:
            if nme:
                v |= FLAGS[nme.replace("AEFFT_", "").replace("F_", "")]
        self.check(self.L.aefft_ctx_set_flags(self.h, v))

    def get_flags(self):
        return int(self.L.aefft_ctx_get_flags(self.h))

    # ---- profiling ----
    def prof_enable(self, on=True):
        self.check(self.L.aefft_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.L.aefft_prof_reset(self.h))

    def prof_read(self):
        out = {}
        for kid in range(self.L.aefft_prof_count()):
            n, ms, by = C.c_long(), C.c_double(), C.c_double()
            self.check(self.L.aefft_prof_read(self.h, kid, C.byref(n), C.byref(ms), C.byref(by)))
            out[self.L.aefft_prof_name(kid).decode()] = dict(launches=n.value, ms=ms.value, bytes=by.value)
        return out


class Net:
    """aefft_net: resident batched autoencoder (autoenc_fft / backprop_fft semantics)."""

    def __init__(self, ctx, D, Nx, Ny, maps, Nk, scale, batch, Nl=None):
        self.ctx, self.L = ctx, ctx.L
        self.D, self.Nx, self.Ny, self.B = D, Nx, Ny, batch
        self.maps = list(maps); self.npairs = len(self.maps)
        self.Nk = [Nk] * self.npairs if isinstance(Nk, int) else list(Nk)
        self.Nl = list(self.Nk) if Nl is None else ([Nl] * self.npairs if isinstance(Nl, int) else list(Nl))
        self.scale = [scale] * self.npairs if isinstance(scale, int) else list(scale)
        arr = lambda v: (C.c_int * self.npairs)(*v)
        self._keep = [arr(self.maps), arr(self.Nk), arr(self.Nl), arr(self.scale)]
        d = NetDesc(D, Nx, Ny, self.npairs, *self._keep, batch)
        h = C.c_void_p()
        ctx.check(self.L.aefft_net_create(ctx.h, C.byref(d), C.byref(h)))
        self.h = h
        # per-pair geometry
        self.dims = []
        dD, nx, ny = D, Nx, Ny
        for l in range(self.npairs):
            nx //= self.scale[l]; ny //= self.scale[l]
            self.dims.append(dict(dD=dD, dM=self.maps[l], Nx=nx, Ny=ny, Nk=self.Nk[l], Nl=self.Nl[l]))
            dD = self.maps[l]

    def close(self):
        if getattr(self, "h", None):
            self.L.aefft_net_destroy(self.h)
            self.h = None

    def __del__(self):
        if _exiting:      # interpreter shutdown: the HIP runtime may already be gone; the process frees everything
            return
        try:
            self.close()
        except Exception:
            pass

    def set_pair(self, l, c, b, f, p):
        a = [np.ascontiguousarray(v, np.float32) for v in (c, b, f, p)]
        g = self.dims[l]
        assert a[0].shape == (g["dM"], g["dD"], g["Nk"], g["Nl"]) and a[2].shape == (g["dD"], g["dM"], g["Nk"], g["Nl"])
        assert a[1].shape == (g["dM"],) and a[3].shape == (g["dD"],)
        self.ctx.check(self.L.aefft_net_set_pair(self.h, l, *[_hptr(v) for v in a]))

    def get_pair(self, l):
        g = self.dims[l]
        c = np.empty((g["dM"], g["dD"], g["Nk"], g["Nl"]), np.float32); f = np.empty((g["dD"], g["dM"], g["Nk"], g["Nl"]), np.float32)
        b = np.empty(g["dM"], np.float32); p = np.empty(g["dD"], np.float32)
        self.ctx.check(self.L.aefft_net_get_pair(self.h, l, _hptr(c), _hptr(b), _hptr(f), _hptr(p)))
        return c, b, f, p

    def store_spectra(self, l):
        g = self.dims[l]
        nyr = g["Ny"] // 2 + 1
        Cs = np.empty((g["dM"], g["dD"], g["Nx"], nyr), np.complex64); Fs = np.empty((g["dD"], g["dM"], g["Nx"], nyr), np.complex64)
        self.ctx.check(self.L.aefft_net_store_spectra(self.h, l, _hptr(Cs), _hptr(Fs)))
        return Cs, Fs

    def load_spectra(self, l, Cs, b, Fs, p):
        a = [np.ascontiguousarray(Cs, np.complex64), np.ascontiguousarray(b, np.float32),
             np.ascontiguousarray(Fs, np.complex64), np.ascontiguousarray(p, np.float32)]
        self.ctx.check(self.L.aefft_net_load_spectra(self.h, l, *[_hptr(v) for v in a]))

    def forward(self, frames, recon=None):
        fn = self.L.aefft_net_forward_u8 if _is_u8(frames) else self.L.aefft_net_forward
        self.ctx.check(fn(self.h, _ptr(frames), _ptr(recon)))
        return recon

    def get_layer(self, layer):
        ch, nx, ny = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.L.aefft_net_get_layer(self.h, layer, None, C.byref(ch), C.byref(nx), C.byref(ny)))
        out = self.ctx.empty(self.B, ch.value, nx.value, ny.value)
        self.ctx.check(self.L.aefft_net_get_layer(self.h, layer, _ptr(out), None, None, None))
        return out

    def get_layers(self):
        """every layer 0..4L of the last forward in one call: list of torch views into one packed buffer"""
        nl = 4 * self.npairs + 1
        offs = (C.c_size_t * (nl + 1))()
        self.ctx.check(self.L.aefft_net_layers_layout(self.h, offs))
        buf = self.ctx.empty(int(offs[nl]))
        self.ctx.check(self.L.aefft_net_get_layers(self.h, _ptr(buf)))
        out = []
        for l in range(nl):
            ch, nx, ny = C.c_int(), C.c_int(), C.c_int()
            self.ctx.check(self.L.aefft_net_get_layer(self.h, l, None, C.byref(ch), C.byref(nx), C.byref(ny)))
            out.append(buf[int(offs[l]):int(offs[l + 1])].view(self.B, ch.value, nx.value, ny.value))
        return out

    def train_pair(self, l, n_iter, del0, maxdiff=0, sym=0):
        mse = np.zeros(n_iter + 1, np.float32)
        self.ctx.check(self.L.aefft_net_train_pair(self.h, l, n_iter, del0, maxdiff, sym, _hptr(mse)))
        return mse

    def set_input_ready(self, on=True):
        """Frames given to step_grad are complete when the call is made: their R2C may run ahead on a side stream."""
        self.ctx.check(self.L.aefft_net_set_input_ready(self.h, 1 if on else 0))

    def step_grad(self, frames, recon=None):
        """frames: float32 [B][D][Nx][Ny], or uint8 of the same shape (8-bit pixels: aefft_net_step_grad_u8, converted by the input transform)."""
        fn = self.L.aefft_net_step_grad_u8 if _is_u8(frames) else self.L.aefft_net_step_grad
        self.ctx.check(fn(self.h, _ptr(frames), _ptr(recon)))

    def grad_buffer(self):
        """torch view of the packed gradient buffer (for torch.distributed.all_reduce)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx.check(self.L.aefft_net_grad_buffer(self.h, C.byref(p), C.byref(n)))
        t = self.ctx.torch
        if not hasattr(self, "_gradview") or self._gradview[0] != p.value:
            iface = {"shape": (n.value,), "typestr": "<f4", "data": (p.value, False), "version": 2, "strides": None}
            holder = type("_Raw", (), {"__cuda_array_interface__": iface})()
            self._gradview = (p.value, t.as_tensor(holder, device=f"cuda:{self.ctx.device}"))
        return self._gradview[1]

    def mse_prev_global(self):
        """torch view of the L floats behind the packed buffer: the global-batch post-update MSE per pair of the step BEFORE the last
        aefft_net_step_apply (the reduced tail of the last all-reduce times grad_scale, saved by step_apply itself; include/aefft.h)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx.check(self.L.aefft_net_grad_buffer(self.h, C.byref(p), C.byref(n)))
        t = self.ctx.torch
        if not hasattr(self, "_prevview") or self._prevview[0] != p.value:
            iface = {"shape": (self.npairs,), "typestr": "<f4", "data": (p.value + 4 * n.value, False), "version": 2, "strides": None}
            holder = type("_Raw", (), {"__cuda_array_interface__": iface})()
            self._prevview = (p.value, t.as_tensor(holder, device=f"cuda:{self.ctx.device}"))
        return self._prevview[1]

    def step_form(self):
        """which form the next training step runs in: "per_frame", "operator" or "operator_chain" (aefft_net_step_form)"""
        return ("per_frame", "operator", "operator_chain")[self.L.aefft_net_step_form(self.h)]

    def step_apply(self, del0, maxdiff=0, sym=0, grad_scale=1.0, mse=None):
        self.ctx.check(self.L.aefft_net_step_apply(self.h, del0, maxdiff, sym, grad_scale, _ptr(mse)))

    def last_mse(self, mse=None):
        """per-pair post-update MSE of the last step_apply (aefft_net_last_mse): sums them now if step_apply(mse=None) left that to the next step"""
        if mse is None:
            mse = self.ctx.empty(self.npairs)
        self.ctx.check(self.L.aefft_net_last_mse(self.h, _ptr(mse)))
        return mse

    def reset_momentum(self):
        self.ctx.check(self.L.aefft_net_reset_momentum(self.h))
