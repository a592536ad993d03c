"""MI355X-native hot path of tud-amr/multi-robot-fabrics: the per-control-step fabric solve and
the Rollout-Fabrics forward simulation as HIP kernels behind a C ABI (include/mrf.h), with a Python
host layer that keeps the reference's planner-facing call surface.

The directory is named `multi-robot-fabrics_amd`; import it as `multi_robot_fabrics_amd`
(the alias module at the repository root)."""
from . import abi, config, leafspec  # noqa: F401

__all__ = ["abi", "config", "leafspec"]
