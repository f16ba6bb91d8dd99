"""CPU-side boundary checks: libaefft.so loads and exports every symbol include/aefft.h declares
(no compute call is made -- there is no GPU here), and the Python prototype table covers them all."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
aefft = importlib.import_module("autoencoder-fft_amd")


def _declared():
    txt = open(os.path.join(ROOT, "include", "aefft.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aefft_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(aefft.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return aefft.lib()


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", aefft.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [n for n in names if n not in exported]
    assert not missing, missing


def test_python_prototypes_cover_header(built):
    assert sorted(aefft.SIGNATURES) == _declared()


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aefft.AefftError):
        aefft.Context()
    import ctypes as C
    h = C.c_void_p()
    assert built.aefft_ctx_create(C.byref(h), 0, None, 1) == aefft.EHIP and not h.value


def test_product_does_not_touch_oracle():
    """the product path must never import / link the checkers under oracle/"""
    pkg = os.path.join(ROOT, "autoencoder-fft_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "np_ref" not in txt and "cpu_ref" not in txt and "oracle/" not in txt.replace("oracle/ ", ""), (dp, fn)
