"""Mirror of the reference's `eagle_mpc.utils` package as far as the hot path's callers use it
(bindings/python/eagle_mpc/utils/): tools (log format), simulator (AerialSimulator: the RK4 plant of the closed-loop
examples), path (EAGLE_MPC_YAML_DIR)."""
from . import path, simulator, tools  # noqa: F401
from .simulator import AerialSimulator  # noqa: F401
from .tools import CallbackLogger, loadLogfile, saveLogfile  # noqa: F401
