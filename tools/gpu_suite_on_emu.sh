#!/bin/bash
# Pre-flight for a change to the HOST code when no GPU is at hand: the `-m gpu` test files over the emulated device layer
# (tests/_emu_plugin.py -> oracle/libnlopt_amd_emu.so).  Everything the GPU suite asserts about drivers, dispatcher, collectives and
# error paths is checked against the same oracle; the HIP kernels are NOT (the emulation replaces them).  Left out: what needs the
# real device or the real library file — the reference's client programs and CLI linked against libnlopt_amd.so, user kernels
# (code objects), the chain kernel's own test, RCCL itself, wall-clock tests and the tests that interrupt a device-resident search from outside (the emulated device is
# synchronous: such a search would never end), the device's own libm check — and the full-size cases (hours on a CPU; ISRES above 2^20 individuals: > 10 min).
#   bash tools/gpu_suite_on_emu.sh [extra pytest args]        (2-3 min with 6 workers; 400+ tests)
cd "$(dirname "$0")/.." || exit 1
make -s -C oracle port emu emudev mockrccl || exit 1
PYTHONPATH=tests NLA_TEST_EMU_DEVICE=1 python -m pytest -p _emu_plugin tests -m gpu -q -p no:cacheprovider -n "${JOBS:-6}" --timeout 600 --tb=line -rf \
    --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_zz_clients.py --deselect tests/test_gpu_testopt_cli.py \
    --deselect tests/test_gpu_cpp_client.py --deselect tests/test_gpu_userobj.py --deselect tests/test_gpu_dropin.py \
    --deselect tests/test_gpu_kernels.py::test_chain_kernel_resolves_the_window_like_the_sequential_statement \
    --deselect tests/test_gpu_chain_resolver.py::test_chain_kernel_with_the_dedicated_resolver \
    --deselect "tests/test_gpu_chain_resolver.py::test_the_resolver_changes_nothing_but_who_advances_the_chain[rastrigin-512-100000-2500]" \
    --deselect "tests/test_gpu_chain_resolver.py::test_the_resolver_changes_nothing_but_who_advances_the_chain[griewank-2048-100000-1500]" \
    --deselect tests/test_gpu_lbfgs.py::test_device_sincos_is_sin_and_cos \
    --deselect tests/test_gpu_isres.py::test_population_above_2pow20_without_constraints \
    --deselect tests/test_gpu_multiproc.py::test_rccl_transport_one_rank \
    --deselect tests/test_gpu_isres.py::test_full_size_config3_parallel_evolve_equals_the_serial_chain \
    --deselect tests/test_gpu_crs.py::test_full_size_invariants_at_the_metric_configuration \
    --deselect "tests/test_gpu_isres.py::test_overlap_mode_changes_nothing[rastrigin-256-50000-42-4-0-kw3]" \
    --deselect tests/test_gpu_stops.py::test_maxtime_stops_the_run \
    --deselect tests/test_gpu_stops.py::test_maxtime_is_observed_inside_a_device_resident_local_search \
    --deselect tests/test_gpu_stops.py::test_force_stop_from_another_thread_ends_a_device_resident_search "$@"
