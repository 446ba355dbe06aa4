"""Import helper: the package directory is named `turbo-metrics_amd/` (not a valid Python identifier),
so it is registered under the module name `turbo_metrics_amd`.

    from tm_pkg import tm      # tm.ffi, tm.TurboMetrics, tm.synth ...
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_ROOT, "turbo-metrics_amd")


def load():
    if "turbo_metrics_amd" in sys.modules:
        return sys.modules["turbo_metrics_amd"]
    spec = importlib.util.spec_from_file_location(
        "turbo_metrics_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["turbo_metrics_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


tm = load()
