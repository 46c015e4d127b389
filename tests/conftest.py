import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "keep_garbage: skip the quiescent teardown (the test is about uncollected GPU objects)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure only)."""
    from oracle import iso_oracle
    iso_oracle.build()
    return iso_oracle


@pytest.fixture
def diag_lib():
    """The DIAGNOSTICS build of the kernel library for the duration of one test (``ops.diagnostics_library()``): fault injection for the
    timeout paths, switches that force a kernel form no shape selects, the experimental forms.  Every other GPU test runs on the product
    build -- lib/libisr_sr.so, which has no ``isrDebug*`` symbol."""
    from isosurfacesuperresolution_amd import ops
    with ops.diagnostics_library() as lib:
        yield lib


@pytest.fixture(autouse=True)
def _quiescent_teardown(request):
    """GPU tests: drain the device and collect this test's garbage (HIP graphs, streams, events, pipelines) at its end, i.e. at a
    quiescent point -- hygiene between test modules, NOT what keeps the library safe: the one place where a collection at the
    wrong moment is fatal (a cyclic collection inside an open stream capture, round 3's abort) is guarded in the library itself
    (``ops.graph_capture``), and ``test_fit_gpu.py::test_capture_survives_uncollected_gpu_garbage`` runs without this teardown."""
    yield
    if request.node.get_closest_marker("gpu") is not None and request.node.get_closest_marker("keep_garbage") is None:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
