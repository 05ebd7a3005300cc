import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "fabric-mod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# This image carries two ROCm runtimes: the system one (7.2, what libfabgpu.so links) and the one bundled with the PyTorch wheel
# (7.0).  The dynamic loader keeps whichever libamdhip64 arrives first; if that is the system one, a later torch.cuda
# initialisation in the same process fails ("No HIP GPUs are available", seen on the GPU box when a test module that only
# uses the C ABI ran before any module that imports torch).  Tests use both, so torch's runtime goes in first.
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
