import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "latent-nerf-test_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """Path of liblnerf_hip.so; (re)built with hipcc when sources changed (no GPU needed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lnerf_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("HIPCC"):
        return mod.build(verbose=False)
    if not os.path.exists(mod.LIB):
        pytest.skip("hipcc not available and library not prebuilt")
    return mod.LIB


@pytest.fixture(scope="session")
def bits_oracle():
    """ctypes handle of the C integer oracle (oracle/bits.c), compiled on demand with gcc."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, "oracle", "libbits_oracle.so")
    src = os.path.join(ROOT, "oracle", "bits.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)
