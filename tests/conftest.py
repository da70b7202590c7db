import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


CONFIGS = {
    # name: (yaml relative to eagle-mpc_amd/data/yaml, dt in ms)  -- BASELINE.json configs 1-4
    "hover": ("hexacopter370/trajectories/hover.yaml", 40),
    "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
    "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
    "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13),
}


@pytest.fixture(scope="session")
def empc():
    import empc_loader
    mod = empc_loader.load()
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mod


@pytest.fixture(scope="session")
def problems(empc):
    """name -> (Trajectory, Problem)"""
    out = {}
    for name, (rel, dt) in CONFIGS.items():
        t = empc.Trajectory()
        t.autoSetup(empc.yaml_path(rel))
        out[name] = (t, t.createProblem(dt, True, "IntegratedActionModelEuler"))
    return out
