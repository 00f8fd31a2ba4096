import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: whole frames / training steps on the host build of the kernels, or child processes hosting the reference "
                                       "(minutes each).  Quicker loop: -m 'not gpu and not slow' (148 tests, ~20 min on 8 cores against ~44 for everything); the driver's -m 'not gpu' runs everything")


SLOW_MODULES = ('test_hipcpu_frame', 'test_reference_dropin', 'test_reference_trainstep', 'test_checkpoint_roundtrip', 'test_dist_cpu', 'test_hipcpu_nn')


def pytest_collection_modifyitems(config, items):
    for it in items:
        if it.module.__name__.rsplit('.', 1)[-1] in SLOW_MODULES:
            it.add_marker(pytest.mark.slow)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _poison_uninitialised_backward_buffers(monkeypatch):
    """The backward allocates the buffers a kernel overwrites completely with torch.empty (sherf_amd/backward_dense.py: Mat.empty).
    Under test they are filled with NaN, so a kernel that leaves part of one unwritten shows up in every gradient check."""
    import torch
    from sherf_amd import backward_dense

    def poisoned(rows, cols, device):
        return backward_dense.Mat(torch.full((int(rows) * int(cols),), float('nan'), dtype=torch.float32, device=device), rows, cols)
    monkeypatch.setattr(backward_dense.Mat, 'empty', staticmethod(poisoned))
