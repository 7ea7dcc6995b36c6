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
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port", "emu", "emudev", "mockrccl"], check=True)
    if os.path.isdir("/root/reference/src"):
        ref = os.path.join(ROOT, "oracle", "_ref")
        if not os.path.exists(os.path.join(ref, "libnlopt_ref.so")):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
        if not os.path.exists(os.path.join(ref, "testopt_amd")) or not os.path.exists(os.path.join(ref, "t_bounded_amd")) \
                or not os.path.exists(os.path.join(ref, "t_tutorial_amd")) or not os.path.exists(os.path.join(ref, "cpp_functor_amd")):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "cpptest"], check=True)


# Collection order of the -m gpu suite: the hot path first — the end-to-end CRS2_LM parity tests and BASELINE's configurations at
# full size are the first 30 tests, then the kernel-level tests, the other algorithms of the path, and the periphery (host
# algorithms, client programs) last — so that with `-x` a surprise in the periphery cannot hide the hot path from whoever reads
# the log (round-2 verdict, item 2).  Files not listed (the CPU suite) keep their alphabetical order in front.
GPU_ORDER = ["test_gpu_crs", "test_gpu_fullsize", "test_gpu_kernels", "test_gpu_crs_windows", "test_gpu_isres", "test_gpu_mlsl", "test_gpu_lbfgs",
             "test_gpu_exact_local", "test_gpu_mma", "test_gpu_esch", "test_gpu_stops", "test_gpu_fixed_dims", "test_gpu_userobj",
             "test_gpu_maximise", "test_gpu_nan", "test_gpu_host_callbacks", "test_gpu_multiproc", "test_gpu_cobyla", "test_gpu_dropin",
             "test_gpu_cpp_client", "test_gpu_testopt_cli", "test_gpu_zz_clients"]


def pytest_collection_modifyitems(session, config, items):
    if os.environ.get("NLA_TEST_KEEP_ORDER"):         # (tools/history/hunt.sh replays the round-2 collection order with it)
        return
    rank = {m: i for i, m in enumerate(GPU_ORDER)}

    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        return rank.get(mod, -1 if not mod.startswith("test_gpu_") else len(GPU_ORDER))
    items.sort(key=key)           # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def L():
    import nlopt_amd
    return nlopt_amd.lib()


def has_gpu():
    import nlopt_amd
    return nlopt_amd.device_count() > 0
