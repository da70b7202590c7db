"""eagle_mpc.utils.path (bindings/python/eagle_mpc/utils/path.py): where the YAML files and robot descriptions live."""
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EAGLE_MPC_YAML_DIR = os.path.join(_PKG, "data", "yaml")
EAGLE_MPC_ROBOT_DATA_DIR = os.path.join(_PKG, "data", "robots")
