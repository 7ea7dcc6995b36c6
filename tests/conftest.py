"""pytest configuration: the `gpu` marker, and one-time builds of the checker (oracle/) and of the
product library (nlopt_amd/lib/libnlopt_amd.so) when they are missing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    lib = os.path.join(ROOT, "nlopt_amd", "lib", "libnlopt_amd.so")
    if not os.path.exists(lib):
        import nlopt_amd
        nlopt_amd.build()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port", "emu", "emudev"], check=True)
    if os.path.isdir("/root/reference/src"):
        ref = os.path.join(ROOT, "oracle", "_ref")
        if not os.path.exists(os.path.join(ref, "libnlopt_ref.so")):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
        if not os.path.exists(os.path.join(ref, "testopt_amd")) or not os.path.exists(os.path.join(ref, "t_bounded_amd")) \
                or not os.path.exists(os.path.join(ref, "t_tutorial_amd")) or not os.path.exists(os.path.join(ref, "cpp_functor_amd")):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "cpptest"], check=True)


@pytest.fixture(scope="session")
def L():
    import nlopt_amd
    return nlopt_amd.lib()


def has_gpu():
    import nlopt_amd
    return nlopt_amd.device_count() > 0
