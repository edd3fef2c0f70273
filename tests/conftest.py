import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The LD1 expansion of a batch's unknown words runs on the device or on the host, whichever is estimated cheaper (csrc/host/engine.cpp, ld1_on_device) — with the
# small batches of the tests that would be the host.  The GPU suite pins it to the DEVICE so that every parity test exercises k_ld1 (k_wm always runs on the device);
# the host walk is covered by the CPU suite and by the device-vs-host A/B test.
os.environ.setdefault("INFX_DEVICE_LOOKUPS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """torch ships its own HIP runtime; when a test also uses torch CUDA tensors (the RCCL exchange-buffer path of the sharded
    phases) torch must initialise the GPU before libinfidex_hip.so pulls in the system runtime, exactly as bench.py does."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
