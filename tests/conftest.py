import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    import torch
    # many-core hosts: torch's CPU pools crawl on tiny ops with hundreds of threads
    torch.set_num_threads(min(8, torch.get_num_threads()))
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def hip():
    """The ctypes binding over libctcasr.so; GPU tests call the C ABI through it."""
    import torch
    from ctc_asr_amd import hip as _hip
    assert torch.cuda.is_available(), 'GPU test selected but no GPU is visible'
    _hip.load()
    return _hip
