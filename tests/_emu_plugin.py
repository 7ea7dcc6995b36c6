"""pytest plugin for tools/gpu_suite_on_emu.sh (`python -m pytest -p _emu_plugin -m gpu ...` with PYTHONPATH=tests): points the package at
oracle/libnlopt_amd_emu.so — the product's C sources linked with the CPU stand-in for the device layer (oracle/emu_device.c) — in the
main process and in every xdist worker, so that the `-m gpu` test files run on a machine without a GPU.  What that checks is the HOST
side of what those tests exercise (drivers, dispatcher, collectives, error paths) against the same oracle assertions; it says nothing
about the HIP kernels, which the emulation replaces.  Test infrastructure only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NLA_TEST_EMU_DEVICE"] = "1"           # tests/_mp_worker.py: the ranks of the multi-process tests load the same library
import nlopt_amd  # noqa: E402

nlopt_amd.LIB_PATH = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
