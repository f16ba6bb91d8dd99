import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library keeps reconstructions below 8 MB on the context stream; the tests' small nets are to take the side-stream path that the benchmark
# shapes take (AEFFT_F_SMALLOVERLAP, process-wide through AEFFT_FLAGS: read once, when the first context is created)
os.environ["AEFFT_FLAGS"] = ",".join(x for x in (os.environ.get("AEFFT_FLAGS", ""), "SMALLOVERLAP") if x)
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def flags(ctx):
    """Set the library's development switches (aefft_ctx_set_flags) for one test; the defaults come back afterwards."""
    def setter(*names):
        ctx.set_flags(*[n for n in names if n])
    yield setter
    ctx.set_flags()
