"""`import nlopt` for the scripts run by tests/test_python_module.py (the reference's own test/t_python.py, test/t_memoize.py
and tests/pyapi/seeded_runs.py): the SWIG-compatible module of this repository under the reference's module name."""
from nlopt_amd.nlopt import *                      # noqa: F401,F403
from nlopt_amd.nlopt import __getattr__            # noqa: F401  (module-level __version__)
