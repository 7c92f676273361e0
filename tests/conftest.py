import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled, unmodified reference (oracle/_ref).  Container only; absent on the GPU box
    unless the prebuilt .so travelled with the snapshot."""
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libnblic_ref.so not built here")
    return Reference()


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("nblic-image-compression_amd")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "manifest.json")) as f:
        manifest = json.load(f)
    streams = np.load(os.path.join(g, "small_streams.npz"), allow_pickle=False)
    return manifest, streams


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    # PyTorch-ROCm bundles its own HIP runtime: when both live in one process torch must
    # initialise first so the library binds to the runtime that is already loaded.
    try:
        import torch
        torch.cuda.init()
    except Exception:
        pass
    ctx = pkg.Context(device=0, n_slots=3, n_coders=3)
    yield ctx
    ctx.close()
