"""Import helper: the package directory is called ``eagle-mpc_amd`` (hyphen), so it is loaded by path under the
module name ``eagle_mpc_amd``.  Also hosts the ctypes binding of the ORACLE for the only callers allowed to use it
(tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "eagle-mpc_amd")


def load():
    name = "eagle_mpc_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
